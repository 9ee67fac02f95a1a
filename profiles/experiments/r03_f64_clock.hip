// Effective shader clock of MI355X under a chip-wide f64 VALU load (the CTM solve phase is such a load): every SIMD runs W waves of
// independent v_fma_f64 (4 issue cycles each, measured in r03_f64_rates); clock = instructions x 4 / (SIMD-seconds).
// Also reads clock64() (shader cycles) against wall_clock64() (constant 100 MHz) inside the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* clk, int iters, double b, double c)
{
    double r[8];
    for (int i = 0; i < 8; ++i) r[i] = 1.0 + i + threadIdx.x * 1e-3;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[j]) : "v"(b), "v"(c));
    }
    const unsigned long long w1 = wall_clock64(), c1 = clock64();
    double s = 0; for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = w1 - w0; clk[2 * blockIdx.x + 1] = c1 - c0; }
}
int main()
{
    for (int wps = 1; wps <= 4; ++wps) {
        const int blocks = 256 * wps, iters = 40000;
        double* out; unsigned long long* clk;
        hipMalloc(&out, sizeof(double) * blocks * 256); hipMalloc(&clk, 16 * blocks);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<blocks, 256>>>(out, clk, 1000, 1.0000001, 1e-9); hipDeviceSynchronize();
        hipEventRecord(e0); k<<<blocks, 256>>>(out, clk, iters, 1.0000001, 1e-9); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double inst_per_simd = (double)wps * iters * 64;      // wave instructions per SIMD
        printf("waves/SIMD %d: %.3f ms, %.1f TFLOP/s f64, issue-limited clock estimate %.3f GHz; in-kernel clock64/wall_clock64(100 MHz): %.3f GHz\n", wps, ms,
               2.0 * 64 * inst_per_simd * 1024 / (ms * 1e-3) / 1e12, inst_per_simd * 4 / (ms * 1e-3) / 1e9, (double)h[1] / ((double)h[0] / 100e6) / 1e9);
        hipFree(out); hipFree(clk);
    }
    return 0;
}
