#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
__global__ void k(const double* x, double* r, double* q, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { r[i] = __builtin_amdgcn_rcp(x[i]); q[i] = __builtin_amdgcn_rsq(x[i]); } }
static int64_t bits(double v) { int64_t b; memcpy(&b, &v, 8); return b; }
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n), r(n), q(n);
    std::mt19937_64 g(7);
    for (int i = 0; i < n; ++i) { double m = 1.0 + (double)(g() >> 11) * (1.0 / 9007199254740992.0); int e = (int)(g() % 600) - 300; x[i] = ldexp(m, e); }
    double *dx, *dr, *dq; (void)hipMalloc(&dx, 8 * n); (void)hipMalloc(&dr, 8 * n); (void)hipMalloc(&dq, 8 * n);
    (void)hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dr, dq, n);
    (void)hipMemcpy(r.data(), dr, 8 * n, hipMemcpyDeviceToHost); (void)hipMemcpy(q.data(), dq, 8 * n, hipMemcpyDeviceToHost);
    long long maxu = 0, maxq = 0; double maxrel = 0, maxrelq = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / (long double)x[i];
        const double ex = (double)t;
        long long u = llabs(bits(r[i]) - bits(ex)); if (u > maxu) maxu = u;
        double rel = fabs((double)(((long double)r[i] - t) / t)); if (rel > maxrel) maxrel = rel;
        const long double tq = 1.0L / sqrtl((long double)x[i]);
        long long uq = llabs(bits(q[i]) - bits((double)tq)); if (uq > maxq) maxq = uq;
        double relq = fabs((double)(((long double)q[i] - tq) / tq)); if (relq > maxrelq) maxrelq = relq;
    }
    printf("v_rcp_f64: max ulp diff %lld, max rel err %.3g (2^%.1f)\n", maxu, maxrel, log2(maxrel));
    printf("v_rsq_f64: max ulp diff %lld, max rel err %.3g (2^%.1f)\n", maxq, maxrelq, log2(maxrelq));
    return 0;
}
