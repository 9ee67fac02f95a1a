// Issue cost, dependent latency and VALU co-execution of v_mfma_f64_16x16x4_f64 on one CU of gfx950 (the dense-row LDA E-step has three
// K x V contractions per 16 documents that could leave the vector pipe -- VERDICT r3 item 4).  One block, W waves; s_memtime-style cycle
// counter around N unrolled instructions.
//   mode 0: every wave issues independent MFMAs (4 accumulators)        mode 1: dependent chain (one accumulator)
//   mode 2: waves 0..3 (one per SIMD) issue MFMAs, waves 4..7 (their SIMD partners) issue independent v_fma_f64: both timed
//   mode 3: ONE wave interleaves 1 MFMA with F independent v_fma_f64 (F = 4, 8, 16): does the vector work hide under the matrix op?
// build: hipcc -O3 --offload-arch=gfx950 -o r04_mfma_f64 r04_mfma_f64.hip ; run: ./r04_mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define ITER 256

template <int MODE, int F>
__global__ void k(double* out, unsigned long long* cyc, double a0, double b0)
{
    const int wid = threadIdx.x >> 6;
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = d4{a0, a0 + i, a0 + 2, a0 + 3};
    double a = a0 + threadIdx.x * 1e-3, b = b0;
    double r[16];
    for (int i = 0; i < 16; ++i) r[i] = a0 + i;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    const bool mf = MODE != 2 || wid < 4;
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 0 || (MODE == 2 && mf)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (F == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[j]) : "v"(b), "v"(a));
                if (F == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(*(float*)&r[j]) : "v"((float)b));
                if (F == 2) asm volatile("v_rcp_f64 %0, %0" : "+v"(r[j]));
                if (F == 3) { int lo = __double2loint(r[j]), hi = __double2hiint(r[j]); asm volatile("v_and_b32 %0, %0, %1" : "+v"(lo) : "v"(hi)); r[j] = __hiloint2double(hi, lo); }
                if (F == 4) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r[j]) : "v"(wid + j));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < F; ++f) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[f & 15]) : "v"(b), "v"(a));
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 16; ++i) s += r[i];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[wid] = t1 - t0;
}

template <int MODE, int F>
void run(const char* name, int waves)
{
    double* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * 64 * 16); hipMalloc(&cyc, 8 * 16);
    for (int rep = 0; rep < 2; ++rep) { k<MODE, F><<<1, 64 * waves>>>(out, cyc, 1.5, 1.0000001); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(16);
    hipMemcpy(h.data(), cyc, 8 * waves, hipMemcpyDeviceToHost);
    if (MODE == 2) {
        unsigned long long m1 = 0, m2 = 0;
        for (int i = 0; i < 4; ++i) m1 = h[i] > m1 ? h[i] : m1;
        for (int i = 4; i < waves; ++i) m2 = h[i] > m2 ? h[i] : m2;
        printf("%-44s MFMA wave: %.2f ticks/mfma   partner wave: %.2f ticks/instr\n", name, (double)m1 / (4.0 * ITER), (double)m2 / (16.0 * ITER));
    } else {
        unsigned long long mx = 0; for (int i = 0; i < waves; ++i) mx = h[i] > mx ? h[i] : mx;
        printf("%-44s waves/SIMD %.2f  ticks per MFMA%s %.2f\n", name, waves / 4.0, MODE == 3 ? " (+ its F v_fma_f64)" : "", (double)mx / (4.0 * ITER));
    }
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0, 0>("mfma_f64_16x16x4 independent", 4); run<0, 0>("mfma_f64_16x16x4 independent", 8); run<0, 0>("mfma_f64_16x16x4 independent", 16);
    run<1, 0>("mfma_f64_16x16x4 dependent (same acc)", 4); run<1, 0>("mfma_f64_16x16x4 dependent (same acc)", 8);
    run<2, 0>("MFMA wave + partner wave: v_fma_f64", 8); run<2, 1>("MFMA wave + partner wave: v_add_f32", 8); run<2, 2>("MFMA wave + partner wave: v_rcp_f64", 8);
    run<2, 3>("MFMA wave + partner wave: v_and_b32", 8); run<2, 4>("MFMA wave + partner wave: v_cvt_f64_i32", 8);
    run<3, 4>("one wave: MFMA + 4 v_fma_f64", 4); run<3, 8>("one wave: MFMA + 8 v_fma_f64", 4); run<3, 16>("one wave: MFMA + 16 v_fma_f64", 4);
    run<3, 8>("two waves/SIMD: MFMA + 8 v_fma_f64", 8); run<3, 16>("two waves/SIMD: MFMA + 16 v_fma_f64", 8);
    return 0;
}
