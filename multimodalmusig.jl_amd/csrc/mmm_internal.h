// mmm_internal.h -- host-side plumbing shared by the translation units of libmmmusig_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mmmusig.h"

struct mmm_p2p;      // p2p.hip: mailbox all-reduce over xGMI

struct mmm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    mmm_p2p* p2p = nullptr;       // mailbox + peer mappings (NULL: not set up)
    bool p2p_on = false;          // all-reduces of <= mailbox capacity go through the p2p kernel
    int num_cu = 256;
    mmm_tuning_opts tune{};       // the caller's choices (mmm_ctx_set_tuning); a handle copies them when it is created
    std::string err;
    char arch[64] = {0};
    // HIP-event spans around the dominant kernel (mmm_ctx_profile_begin/end)
    bool profiling = false;
    int prof_repeat = 1;          // launches of the dominant kernel inside each profiled span (differential timing)
    int prof_phase = 0;           // which launches of a pass the spans bracket (mmm_ctx_profile_select; 0 = the dominant kernel)
    std::vector<hipEvent_t> ev;   // pairs: ev[2i] start, ev[2i+1] stop
    std::vector<int> ev_phase;    // phase of pair i (prof_phase = kProfAll brackets every phase)
    size_t ev_used = 0;
    // pipelined fits: two pinned 64-byte slots + events for in-stream snapshots of a model's control block, so that the host
    // can look at chunk i's stop flag while chunk i+1 is already running (lazily created)
    void* pin_ctl = nullptr;
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    // models created on this context and still alive.  mmm_ctx_destroy with live models only marks the context; the last
    // mmm_*_destroy then releases it -- a garbage-collected host (Julia finalizers run in no particular order) may destroy
    // the context before its models without a use-after-free.
    // Finalizers may run on any thread, hence atomics; the communicator and the mailboxes are released at mmm_ctx_destroy itself (peers
    // and the runtime are still up then), only the stream, the events and the memory that models still reference wait for the last model.
    // side stream of the CTM fit passes: work that only depends on the theta phase (reduction of the gamma statistics, topic M-step) runs
    // beside the solve phase and is joined before the log-likelihood launch (lazily created; always joined within the pass)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::atomic<int> refs{1};                 // live models + 1 for the owner (dropped by mmm_ctx_destroy); the last one out tears down
    std::atomic<bool> closing{false};         // mmm_ctx_destroy has run
    std::atomic<bool> closed_multi{false};    // ... on a context that had a communicator: surviving models must not be used any more
};

// CU count the launch geometry is derived from: the device's, or the pinned one (mmm_tuning_opts.geometry_cus)
inline int mmm_geo_cus(const mmm_ctx* ctx) { return ctx->tune.geometry_cus > 0 ? ctx->tune.geometry_cus : ctx->num_cu; }
inline bool mmm_off(const mmm_tuning_opts& t, unsigned bit) { return (t.disable & bit) != 0; }

void mmm_ctx_model_created(mmm_ctx* ctx);
void mmm_ctx_model_destroyed(mmm_ctx* ctx);      // may delete ctx
int mmm_ctx_usable(mmm_ctx* ctx, const char* what);   // error once a multi-rank context has been destroyed under a live model

// RAII span: records an event pair around a launch while profiling is on
struct ProfSpan {
    mmm_ctx* ctx; bool on;
    explicit ProfSpan(mmm_ctx* c, int phase = 0) : ctx(c), on(c->profiling && (c->prof_phase == phase || c->prof_phase == 8)) {
        if (!on) return;
        if (ctx->ev_phase.size() < ctx->ev_used / 2 + 1) ctx->ev_phase.resize(ctx->ev_used / 2 + 1);
        ctx->ev_phase[ctx->ev_used / 2] = phase;
        if (ctx->ev_used + 2 > ctx->ev.size()) {
            for (int i = 0; i < 2; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) { on = false; return; } ctx->ev.push_back(e); }
        }
        (void)hipEventRecord(ctx->ev[ctx->ev_used], ctx->stream);
    }
    ~ProfSpan() { if (on) { (void)hipEventRecord(ctx->ev[ctx->ev_used + 1], ctx->stream); ctx->ev_used += 2; } }
};

extern thread_local std::string g_mmm_create_error;

inline int mmm_fail(mmm_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (ctx) ctx->err = buf; else g_mmm_create_error = buf;
    return code;
}

#define MMM_HIP(ctx, call)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return mmm_fail((ctx), MMM_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call,      \
                            hipGetErrorString(e_));                                                \
    } while (0)

#define MMM_NCCL(ctx, call)                                                                        \
    do {                                                                                           \
        ncclResult_t r_ = (call);                                                                  \
        if (r_ != ncclSuccess)                                                                     \
            return mmm_fail((ctx), MMM_ERR_RCCL, "%s:%d %s -> %s", __FILE__, __LINE__, #call,     \
                            ncclGetErrorString(r_));                                               \
    } while (0)

#define MMM_CHECK(ctx, cond, ...)                                                                  \
    do { if (!(cond)) return mmm_fail((ctx), MMM_ERR_ARG, __VA_ARGS__); } while (0)

#define MMM_LAUNCH_CHECK(ctx) MMM_HIP(ctx, hipGetLastError())

// device buffer with RAII; all model state lives in these
template <typename T>
struct DevBuf {
    T* p = nullptr; size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) {
        if (p) { (void)hipFree(p); p = nullptr; }
        n = count;
        return hipMalloc((void**)&p, (count ? count : 1) * sizeof(T));
    }
    void swap(DevBuf& o) { std::swap(p, o.p); std::swap(n, o.n); }
};

// ---- xGMI mailbox all-reduce (p2p.hip): argument block and the two device-side halves, shared with kernels that fold the
// exchange into their own epilogue / prologue (lda.hip) --------------------------------------------------------------
constexpr int kP2PMaxRanks = 16;
constexpr size_t kP2PCap = 8192;            // doubles per call (LDA: 961; CTM cfg 4: 2,450)

struct P2PArgs {
    // mailbox base of every rank (peer[rank] = the local one): a table in device memory -- an array inside this by-value
    // argument block would be indexed dynamically, which sends the whole block (and the kernel's other arguments) to scratch
    unsigned long long* const* peer;
    int nranks, rank;
    size_t cap;
    int* err;                                 // device word: sequence number of a call that timed out (0 = none)
    unsigned long long timeout_ticks;         // s_memrealtime ticks (100 MHz)
};

#ifdef __HIPCC__
// store `mine` as element e of call `seq` into every peer's mailbox (two 8-byte words, each tagged with seq)
__device__ __forceinline__ void p2p_send(const P2PArgs& a, unsigned int seq, int e, double mine)
{
    const size_t slot = seq & 1u;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(mine);
    const unsigned long long tag = (unsigned long long)seq << 32;
    const unsigned long long w0 = (bits & 0xffffffffull) | tag, w1 = (bits >> 32) | tag;
    for (int p = 0; p < a.nranks; ++p) {
        if (p == a.rank) continue;
        unsigned long long* dst = a.peer[p] + ((slot * a.nranks + a.rank) * a.cap + e) * 2;
        // one 16-byte write-through store per cell instead of two 8-byte ones: half the xGMI write transactions.  It need not be
        // atomic -- each 8-byte half carries its own tag, and a reader accepts a cell only when both tags match
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 cell = {(unsigned int)w0, (unsigned int)(w0 >> 32), (unsigned int)w1, (unsigned int)(w1 >> 32)};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(cell) : "memory");
    }
}

// element e of call `seq`: own value + the peers' contributions from the local mailbox, summed in rank order.  All pending
// cells are requested together in each polling round (a round costs one memory latency, not one per peer).
__device__ __forceinline__ double p2p_recv_sum(const P2PArgs& a, unsigned int seq, int e, double mine)
{
    const size_t slot = seq & 1u;
    const int n = a.nranks, me = a.rank;
    const unsigned long long* base = a.peer[me] + (slot * n * a.cap + e) * 2;      // rank r's cell: base + r * cap * 2
    const size_t rstride = a.cap * 2;
    unsigned long long w0[kP2PMaxRanks], w1[kP2PMaxRanks];
    unsigned int pending = ((n >= 32 ? 0xffffffffu : ((1u << n) - 1u))) & ~(1u << me);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool failed = false;
    while (pending) {
#pragma unroll
        for (int r = 0; r < kP2PMaxRanks; ++r)
            if ((pending >> r) & 1u) {
                w0[r] = __hip_atomic_load(base + r * rstride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                w1[r] = __hip_atomic_load(base + r * rstride + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
#pragma unroll
        for (int r = 0; r < kP2PMaxRanks; ++r)
            if (((pending >> r) & 1u) && (unsigned int)(w0[r] >> 32) == seq && (unsigned int)(w1[r] >> 32) == seq) pending &= ~(1u << r);
        if (pending) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) { failed = true; break; }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    double sum = 0.0;
#pragma unroll
    for (int r = 0; r < kP2PMaxRanks; ++r)
        if (r < n) sum += (r == me) ? mine : __longlong_as_double((long long)((w0[r] & 0xffffffffull) | (w1[r] << 32)));
    if (failed) atomicExch(a.err, (int)seq);
    return sum;
}
#endif

// true when mmm_allreduce_sum really communicates.  A one-rank communicator is only exercised on request
// (MMM_FORCE_RCCL=1: lets a single-GPU box run the RCCL path)
inline bool mmm_comm_active(const mmm_ctx* ctx)
{
    if (ctx->nranks > 1 && (ctx->p2p_on || ctx->comm)) return true;
    if (!ctx->comm) return false;
    static const bool force = getenv("MMM_FORCE_RCCL") != nullptr;
    return force;
}

// sum-all-reduce of a packed double buffer across the ranks of ctx, on the ctx stream (no-op for a single rank): the p2p
// mailbox kernel when it is set up and the payload fits, ncclAllReduce otherwise (p2p.hip)
int mmm_allreduce_sum(mmm_ctx* ctx, double* dev, size_t count);
int mmm_p2p_setup_over_rccl(mmm_ctx* ctx);
int mmm_p2p_check(mmm_ctx* ctx);
// for kernels that fold the exchange in: the argument block and a fresh sequence number (false: p2p not in use / too large)
bool mmm_p2p_begin(mmm_ctx* ctx, size_t count, P2PArgs* args, unsigned int* seq);
void mmm_p2p_release(mmm_ctx* ctx);
