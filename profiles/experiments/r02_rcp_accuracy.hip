// How accurate is v_rcp_f64 and its Newton refinements on gfx950?  (for dev_rcp in csrc/dev_math.h)
// hipcc --offload-arch=gfx950 -O2 r02_rcp_accuracy.hip -o rcp_accuracy && ./rcp_accuracy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = x[i];
    double r = __builtin_amdgcn_rcp(a);
    r0[i] = r;
    r = fma(fma(-a, r, 1.0), r, r);
    r1[i] = r;
    r = fma(fma(-a, r, 1.0), r, r);
    r2[i] = r;
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> h(n), o0(n), o1(n), o2(n);
    std::mt19937_64 g(7);
    std::uniform_real_distribution<double> u(-30.0, 30.0), m(1.0, 2.0);
    for (int i = 0; i < n; ++i) h[i] = std::ldexp(m(g), (int)u(g));
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    hipMemcpy(o0.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(o1.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(o2.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0; long bad1 = 0, bad2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / (long double)h[i];
        const double c = (double)t;
        e0 = std::fmax(e0, (double)fabsl(((long double)o0[i] - t) / t));
        e1 = std::fmax(e1, (double)fabsl(((long double)o1[i] - t) / t));
        e2 = std::fmax(e2, (double)fabsl(((long double)o2[i] - t) / t));
        bad1 += o1[i] != c; bad2 += o2[i] != c;
    }
    printf("max relative error: v_rcp_f64 %.3e, +1 Newton %.3e (%ld of %d not correctly rounded), +2 Newton %.3e (%ld not correctly rounded)\n", e0, e1, bad1, n, e2, bad2);
    return 0;
}
