// micro-benchmark: cost of a grid-wide barrier among 250 co-resident blocks on MI355X (agent-scope atomic counter + spin),
// with and without a 7.7 KB write-through (sc1) store per block before the barrier and a strided read after it
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(640) void k_bar(unsigned int* counter, double* buf, int nbar, int payload, unsigned long long timeout)
{
    const int nb = gridDim.x;
    unsigned int target = 0;
    double acc = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < nbar; ++it) {
        if (payload) {      // every block publishes 960 doubles, write-through
            for (int i = threadIdx.x; i < 960; i += blockDim.x)
                __hip_atomic_store(&buf[(size_t)blockIdx.x * 960 + i], (double)(it + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        target += nb;
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (__builtin_amdgcn_s_memrealtime() - t0 > timeout) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (payload) {      // every block reads its 4-entry slice of all partials (the reduce step's access pattern)
            for (int s = threadIdx.x; s < nb; s += blockDim.x)
                for (int q = 0; q < 4; ++q) acc += __hip_atomic_load(&buf[(size_t)s * 960 + (blockIdx.x * 4 + q) % 960], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (acc == 12345.678) buf[0] = acc;
}
int main()
{
    unsigned int* c; double* buf;
    hipMalloc(&c, 4); hipMalloc(&buf, sizeof(double) * 250 * 960);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int payload = 0; payload < 2; ++payload)
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(c, 0, 4);
            const int nbar = 2000;
            void* args[] = {&c, &buf, (void*)&nbar, (void*)&payload, nullptr};
            unsigned long long to = 300000000ull; args[4] = &to;
            hipEventRecord(e0);
            hipError_t e = hipLaunchCooperativeKernel((void*)k_bar, dim3(250), dim3(640), args, 0, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("payload=%d launch=%s  %.3f us per barrier round\n", payload, hipGetErrorString(e), ms * 1e3 / nbar);
        }
    return 0;
}
