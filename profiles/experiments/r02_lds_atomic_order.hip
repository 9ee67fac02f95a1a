// r02_lds_atomic_order.hip -- in which order does ds_add_f64 apply the lanes of ONE wave instruction that hit the same
// LDS address?  The CTM theta phase adds theta*n of the 2-4 documents of a wave into one slab with a single instruction per
// topic; documents that hold the same term in the same lane position collide.  The order decides the bits of the sum.
// Build: hipcc --offload-arch=gfx950 -O2 r02_lds_atomic_order.hip -o build/lds_atomic_order ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// pattern: lane -> slot; every lane adds val[lane] to slab[slot[lane]] with ONE instruction
__global__ void k_probe(const double* val, const int* slot, int nslot, double* out, double* old_out)
{
    __shared__ double slab[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < nslot; i += 64) slab[i] = 0.0;
    __syncthreads();
    if (slot[lane] >= 0) unsafeAtomicAdd(&slab[slot[lane]], val[lane]);          // non-returning: ds_add_f64
    __syncthreads();
    for (int i = lane; i < nslot; i += 64) out[i] = slab[i];
    __syncthreads();
    for (int i = lane; i < nslot; i += 64) slab[i] = 0.0;
    __syncthreads();
    double o = -1.0;
    if (slot[lane] >= 0) o = atomicAdd(&slab[slot[lane]], val[lane]);            // returning: ds_add_rtn_f64
    old_out[lane] = o;
}

static double rnd() { return (double)rand() / RAND_MAX; }

int main()
{
    double *dval, *dout, *dold; int* dslot;
    hipMalloc(&dval, 64 * 8); hipMalloc(&dout, 256 * 8); hipMalloc(&dold, 64 * 8); hipMalloc(&dslot, 64 * 4);
    int bad_asc = 0, bad_desc = 0, ncase = 0;
    srand(7);
    for (int rep = 0; rep < 400; ++rep) {
        double val[64]; int slot[64];
        const int kind = rep % 4;
        int nslot = 16;
        for (int l = 0; l < 64; ++l) {
            // magnitudes spread over 2^-30 .. 2^30 so that every order gives different bits
            val[l] = (rnd() - 0.3) * ldexp(1.0, (rand() % 60) - 30);
            if (kind == 0) slot[l] = 0;                                   // all 64 lanes on one address
            else if (kind == 1) slot[l] = l % 16;                         // 4 documents x 16 lanes, same term per position
            else if (kind == 2) slot[l] = (l % 32) / 2;                   // 2 documents x 32 lanes, pairs collide too
            else slot[l] = (rand() % 5 == 0) ? -1 : rand() % 16;          // random with inactive lanes
        }
        hipMemcpy(dval, val, sizeof val, hipMemcpyHostToDevice);
        hipMemcpy(dslot, slot, sizeof slot, hipMemcpyHostToDevice);
        k_probe<<<1, 64>>>(dval, dslot, nslot, dout, dold);
        double out[256], old[64];
        hipMemcpy(out, dout, nslot * 8, hipMemcpyDeviceToHost);
        hipMemcpy(old, dold, sizeof old, hipMemcpyDeviceToHost);
        double asc[16] = {0}, desc[16] = {0};
        for (int l = 0; l < 64; ++l) if (slot[l] >= 0) asc[slot[l]] += val[l];
        for (int l = 63; l >= 0; --l) if (slot[l] >= 0) desc[slot[l]] += val[l];
        for (int s = 0; s < nslot; ++s) {
            ++ncase;
            if (memcmp(&out[s], &asc[s], 8)) ++bad_asc;
            if (memcmp(&out[s], &desc[s], 8)) ++bad_desc;
        }
        // returning atomics: old values must be the ascending prefix sums
        double run[16] = {0};
        int bad_rtn = 0;
        for (int l = 0; l < 64; ++l) if (slot[l] >= 0) { if (memcmp(&old[l], &run[slot[l]], 8)) ++bad_rtn; run[slot[l]] += val[l]; }
        if (rep < 4) printf("kind %d: returning-atomic prefix mismatches vs ascending lane order: %d\n", kind, bad_rtn);
    }
    printf("slots checked %d: differ from ascending-lane sum %d, differ from descending-lane sum %d\n", ncase, bad_asc, bad_desc);
    printf(bad_asc == 0 ? "RESULT: ds_add_f64 applies same-address lanes in ASCENDING lane order\n" : "RESULT: not ascending\n");
    return 0;
}
