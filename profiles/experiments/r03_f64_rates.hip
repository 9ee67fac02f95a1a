// Issue cost and dependent latency of the f64 vector instructions the CTM solve phase is made of, on one SIMD of gfx950.
// One block of 64 x W threads (W waves on ONE CU -> W/4 per SIMD when W is a multiple of 4), s_memtime around N unrolled instructions.
// build: hipcc -O3 --offload-arch=gfx950 -o r03_f64_rates r03_f64_rates.hip ; run: ./r03_f64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 64
#define ITER 64
#define S1(x) x
#define S4(x) x x x x
#define S16(x) S4(x) S4(x) S4(x) S4(x)

template <int OP, bool DEP>
__global__ void k(double* out, unsigned long long* cyc, double a0, double b0)
{
    double r[8];
    for (int i = 0; i < 8; ++i) r[i] = a0 + i + threadIdx.x * 1e-3;
    double b = b0, c = b0 * 0.5;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < REP / 8; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                double& x = DEP ? r[0] : r[j];
                if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
                if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(b));
                if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b));
                if (OP == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
                if (OP == 4) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));
                if (OP == 5) asm volatile("v_sqrt_f64 %0, %0" : "+v"(x));
                if (OP == 6) { int lo = __double2loint(x); asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(lo)); x = __hiloint2double(__double2hiint(x), lo); }
                if (OP == 7) { int lo = __double2loint(x), hi = __double2hiint(x); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(hi)); x = __hiloint2double(hi, lo); }
                if (OP == 8) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
                if (OP == 9) { int lo = __double2loint(x); lo = __builtin_amdgcn_ds_bpermute(threadIdx.x * 4 ^ 64, lo); x = __hiloint2double(__double2hiint(x), lo); }
                if (OP == 10) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(x));
                if (OP == 11) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(b));
                if (OP == 12) asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(x), "v"(b) : "vcc");
                if (OP == 13) asm volatile("v_add_f32 %0, %0, %1" : "+v"(*(float*)&x) : "v"((float)b));
                if (OP == 14) asm volatile("v_mov_b64 %0, %1" : "+v"(x) : "v"(b));
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP, bool DEP>
void run(const char* name, int waves)
{
    double* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * 64 * 16); hipMalloc(&cyc, 8 * 16);
    k<OP, DEP><<<1, 64 * waves>>>(out, cyc, 1.5, 1.0000001);
    hipDeviceSynchronize();
    k<OP, DEP><<<1, 64 * waves>>>(out, cyc, 1.5, 1.0000001);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(16);
    hipMemcpy(h.data(), cyc, 8 * waves, hipMemcpyDeviceToHost);
    unsigned long long mx = 0; for (int i = 0; i < waves; ++i) mx = h[i] > mx ? h[i] : mx;
    // s_memtime / readcyclecounter on gfx9 counts at 100 MHz (REFCLK)? report raw ticks per instruction; calibrate with v_add_f32
    printf("%-28s waves/SIMD %.2f  %s  ticks/instr/wave %.4f\n", name, waves / 4.0, DEP ? "dependent  " : "independent", (double)mx / (REP * ITER));
    hipFree(out); hipFree(cyc);
}

#define RUNALL(OP, name) run<OP, false>(name, 4); run<OP, true>(name, 4); run<OP, false>(name, 8); run<OP, false>(name, 16);

int main()
{
    RUNALL(13, "v_add_f32 (calibration)")
    RUNALL(0, "v_fma_f64") RUNALL(8, "v_fmac_f64") RUNALL(1, "v_add_f64") RUNALL(2, "v_mul_f64") RUNALL(3, "v_rcp_f64") RUNALL(4, "v_rsq_f64") RUNALL(5, "v_sqrt_f64")
    RUNALL(6, "v_mov_b32_dpp") RUNALL(7, "v_cndmask_b32") RUNALL(9, "ds_bpermute_b32") RUNALL(10, "v_ldexp_f64") RUNALL(11, "v_max_f64") RUNALL(12, "v_cmp_lt_f64") RUNALL(14, "v_mov_b64")
    return 0;
}
