// p2p.hip -- the per-iteration all-reduce of the packed sufficient statistics, over xGMI without a collective library.
//
// The payload is 1-20 KB of doubles once per EM iteration: latency is everything, bandwidth nothing.  RCCL's ring costs
// tens of microseconds at this size, i.e. more than the whole 28 us iteration it sits in.  Here every rank owns a MAILBOX in
// fine-grained (uncached, peer-visible) device memory, [2 slots][nranks][cap] cells of 16 bytes; one kernel per call
//   1. stores its own contribution into every peer's mailbox, directly over xGMI, and
//   2. polls its own mailbox for the peers' contributions and sums them in RANK ORDER (same bits on every rank, deterministic).
// A cell carries one double as two 8-byte words {low half | seq} {high half | seq}: each word is written whole, so a cell is
// complete exactly when both words show the sequence number of this call -- no fence, no flag, no ordering requirement
// (the "LL" idea of collective libraries).  A sender writes a cell with ONE 16-byte write-through store (half the xGMI write
// transactions of two 8-byte ones); should the fabric split it, the reader simply sees mismatching tags for a moment.  seq is a per-context call counter (identical on all ranks, which
// issue the same calls in the same order); slot = seq & 1 suffices because a rank cannot start call s+2 before every rank
// has finished reading call s (it needs their s+1 contributions, which they send after their call-s kernel has ended).
// That argument needs EVERY sequence number to be a real exchange.  The LDA kernels that fold the exchange into their own
// work skip it on the no-op passes after the device-side stopping rule has fired (the host has already counted those launches):
// they still send and receive element 0 of that sequence number (value unused), so no rank can run ahead through skipped numbers
// and overwrite a slot a lagging peer is still reading (lda.hip: lda_reduce_block, k_lda_mstep, k_lda_reduce_ll_mstep).
// Every poll loop has a wall-clock exit (default 20 s): on expiry the kernel records the failure and ends, and the next
// host synchronisation point reports it -- a lost peer cannot hang the GPU.
//
// Set-up is transport-agnostic: mmm_p2p_local_handle / mmm_p2p_attach exchange 64-byte IPC handles by whatever the host has.
// mmm_comm_init_rank does it over the RCCL communicator, then runs a self-test against known sums; if anything fails on any
// rank, all ranks keep using ncclAllReduce.
#include "mmm_internal.h"

namespace {

__global__ __launch_bounds__(256) void k_p2p_allreduce(P2PArgs a, double* buf, int count, unsigned int seq)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    const double mine = buf[e];
    p2p_send(a, seq, e, mine);
    buf[e] = p2p_recv_sum(a, seq, e, mine);
}

} // namespace

struct mmm_p2p {
    P2PArgs args{};
    unsigned long long* peer_h[kP2PMaxRanks] = {nullptr};     // host copy of the table args.peer points to
    unsigned long long** peer_d = nullptr;
    void* local = nullptr;
    void* opened[kP2PMaxRanks] = {nullptr};
    int* err = nullptr;
    unsigned int seq = 0;
    size_t bytes = 0;
};

static size_t p2p_bytes(int nranks) { return sizeof(unsigned long long) * 2 * 2 * (size_t)nranks * kP2PCap; }

// the mailbox has to exist before its handle can be handed out; nranks is fixed at that point
static int p2p_alloc(mmm_ctx* ctx, int nranks)
{
    if (ctx->p2p) return MMM_OK;
    MMM_CHECK(ctx, nranks >= 1 && nranks <= kP2PMaxRanks, "p2p: nranks %d not in 1..%d", nranks, kP2PMaxRanks);
    MMM_HIP(ctx, hipSetDevice(ctx->device));
    mmm_p2p* p = new mmm_p2p();
    p->bytes = p2p_bytes(nranks);
    hipError_t e = hipExtMallocWithFlags(&p->local, p->bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { delete p; return mmm_fail(ctx, MMM_ERR_HIP, "p2p: fine-grained allocation of %zu bytes: %s", p->bytes, hipGetErrorString(e)); }
    e = hipMalloc((void**)&p->err, sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&p->peer_d, sizeof(unsigned long long*) * kP2PMaxRanks);
    if (e != hipSuccess) { (void)hipFree(p->local); if (p->err) (void)hipFree(p->err); delete p; return mmm_fail(ctx, MMM_ERR_HIP, "p2p: %s", hipGetErrorString(e)); }
    MMM_HIP(ctx, hipMemset(p->peer_d, 0, sizeof(unsigned long long*) * kP2PMaxRanks));
    p->args.peer = p->peer_d;
    MMM_HIP(ctx, hipMemset(p->local, 0, p->bytes));
    MMM_HIP(ctx, hipMemset(p->err, 0, sizeof(int)));
    MMM_HIP(ctx, hipDeviceSynchronize());
    p->args.nranks = nranks; p->args.cap = kP2PCap; p->args.err = p->err;
    double secs = 20.0;      // generous: a peer may be loading code objects or be descheduled at start-up
    if (const char* s = getenv("MMM_P2P_TIMEOUT_S")) secs = std::max(0.001, atof(s));
    p->args.timeout_ticks = (unsigned long long)(secs * 1e8);
    ctx->p2p = p;
    return MMM_OK;
}

void mmm_p2p_release(mmm_ctx* ctx)
{
    mmm_p2p* p = ctx->p2p;
    if (!p) return;
    (void)hipSetDevice(ctx->device);
    for (int r = 0; r < kP2PMaxRanks; ++r) if (p->opened[r]) (void)hipIpcCloseMemHandle(p->opened[r]);
    if (p->local) (void)hipFree(p->local);
    if (p->err) (void)hipFree(p->err);
    if (p->peer_d) (void)hipFree(p->peer_d);
    delete p;
    ctx->p2p = nullptr; ctx->p2p_on = false;
}

// a timed-out call leaves its sequence number in the error word; called wherever the host synchronises anyway
int mmm_p2p_check(mmm_ctx* ctx)
{
    if (!ctx->p2p || !ctx->p2p_on) return MMM_OK;
    int h = 0;
    MMM_HIP(ctx, hipMemcpyAsync(&h, ctx->p2p->err, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h) return mmm_fail(ctx, MMM_ERR_RCCL, "p2p all-reduce #%d timed out waiting for a peer (rank %d of %d)", h, ctx->rank, ctx->nranks);
    return MMM_OK;
}

static int p2p_launch(mmm_ctx* ctx, double* dev, size_t count)
{
    mmm_p2p* p = ctx->p2p;
    const unsigned int seq = ++p->seq;
    hipLaunchKernelGGL(k_p2p_allreduce, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, p->args, dev, (int)count, seq);
    MMM_LAUNCH_CHECK(ctx);
    return MMM_OK;
}

bool mmm_p2p_begin(mmm_ctx* ctx, size_t count, P2PArgs* args, unsigned int* seq)
{
    if (!ctx->p2p || !ctx->p2p_on || count > kP2PCap) return false;
    *args = ctx->p2p->args;
    *seq = ++ctx->p2p->seq;
    return true;
}

int mmm_allreduce_sum(mmm_ctx* ctx, double* dev, size_t count)
{
    if (!mmm_comm_active(ctx) || count == 0) return MMM_OK;
    if (ctx->p2p_on && count <= kP2PCap) return p2p_launch(ctx, dev, count);
    MMM_CHECK(ctx, ctx->comm != nullptr, "all-reduce of %zu doubles: no RCCL communicator and the payload exceeds the p2p mailbox (%zu)", count, kP2PCap);
    MMM_NCCL(ctx, ncclAllReduce(dev, dev, count, ncclDouble, ncclSum, ctx->comm, ctx->stream));
    return MMM_OK;
}

// known-answer rehearsal: small integers (exact in any summation order), 160 calls with the payload sizes of the real
// path (1 .. the mailbox capacity), so both slots are reused many times before the first real call
static int p2p_selftest(mmm_ctx* ctx, bool* ok)
{
    *ok = false;
    const int n = ctx->nranks, me = ctx->rank, rounds = 160;
    static const int sizes[] = {1, 2, 961, 2450, 17, (int)kP2PCap, 333, 1500};
    DevBuf<double> buf;
    MMM_HIP(ctx, buf.alloc(kP2PCap));
    std::vector<double> h(kP2PCap);
    bool good = true;
    // the first call may wait for a peer that is still attaching (full time limit); once every rank has answered one call, a healthy
    // exchange takes microseconds -- the later calls get 2 s, so that a node where the mailboxes do not work falls back to RCCL quickly
    const unsigned long long full_ticks = ctx->p2p->args.timeout_ticks;
    for (int it = 0; it < rounds && good; ++it) {
        if (it == 1) ctx->p2p->args.timeout_ticks = std::min<unsigned long long>(full_ticks, 200000000ull);
        const int count = sizes[it % 8];
        for (int e = 0; e < count; ++e) h[e] = (double)((me + 1) * (e % 7 + 1) + it);
        MMM_HIP(ctx, hipMemcpyAsync(buf.p, h.data(), sizeof(double) * count, hipMemcpyHostToDevice, ctx->stream));
        int rc = p2p_launch(ctx, buf.p, count);
        if (rc) return rc;
        MMM_HIP(ctx, hipMemcpyAsync(h.data(), buf.p, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
        MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int e = 0; e < count; ++e) {
            const double want = (double)(n * (n + 1) / 2 * (e % 7 + 1) + n * it);
            if (h[e] != want) { good = false; break; }
        }
    }
    ctx->p2p->args.timeout_ticks = full_ticks;
    int herr = 0;
    MMM_HIP(ctx, hipMemcpy(&herr, ctx->p2p->err, sizeof herr, hipMemcpyDeviceToHost));
    if (herr) { good = false; MMM_HIP(ctx, hipMemset(ctx->p2p->err, 0, sizeof(int))); }
    *ok = good;
    return MMM_OK;
}

extern "C" {

int mmm_p2p_local_handle(mmm_ctx* ctx, int nranks, char out[MMM_P2P_HANDLE_BYTES])
{
    static_assert(sizeof(hipIpcMemHandle_t) <= MMM_P2P_HANDLE_BYTES, "hipIpcMemHandle_t larger than the ABI slot");
    if (!ctx || !out) return MMM_ERR_ARG;
    int rc = p2p_alloc(ctx, nranks);
    if (rc) return rc;
    hipIpcMemHandle_t h;
    MMM_HIP(ctx, hipIpcGetMemHandle(&h, ctx->p2p->local));
    memset(out, 0, MMM_P2P_HANDLE_BYTES);
    memcpy(out, &h, sizeof h);
    return MMM_OK;
}

int mmm_p2p_attach(mmm_ctx* ctx, int nranks, int rank, const char* handles)
{
    if (!ctx || !handles) return MMM_ERR_ARG;
    MMM_CHECK(ctx, ctx->p2p && ctx->p2p->args.nranks == nranks, "mmm_p2p_attach: call mmm_p2p_local_handle(ctx, %d, ...) first", nranks);
    MMM_CHECK(ctx, rank >= 0 && rank < nranks, "mmm_p2p_attach: rank %d out of range", rank);
    MMM_CHECK(ctx, ctx->comm == nullptr || (ctx->nranks == nranks && ctx->rank == rank), "mmm_p2p_attach: rank/nranks differ from the RCCL communicator's");
    MMM_HIP(ctx, hipSetDevice(ctx->device));
    mmm_p2p* p = ctx->p2p;
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) { p->peer_h[r] = (unsigned long long*)p->local; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)r * MMM_P2P_HANDLE_BYTES, sizeof h);
        void* q = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return mmm_fail(ctx, MMM_ERR_HIP, "mmm_p2p_attach: hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e));
        p->opened[r] = q; p->peer_h[r] = (unsigned long long*)q;
    }
    MMM_HIP(ctx, hipMemcpy(p->peer_d, p->peer_h, sizeof p->peer_h, hipMemcpyHostToDevice));
    p->args.rank = rank;
    ctx->nranks = nranks; ctx->rank = rank;
    ctx->p2p_on = true;
    return MMM_OK;
}

// every rank calls this after mmm_p2p_attach (it communicates): 0/1 in *ok; a failed rehearsal switches the path off locally --
// the caller must make the decision unanimous (mmm_comm_init_rank does, with an RCCL min-reduction)
int mmm_p2p_selftest(mmm_ctx* ctx, int* ok)
{
    if (!ctx || !ok) return MMM_ERR_ARG;
    MMM_CHECK(ctx, ctx->p2p && ctx->p2p_on, "mmm_p2p_selftest: not attached");
    bool good = false;
    int rc = p2p_selftest(ctx, &good);
    if (rc) return rc;
    *ok = good ? 1 : 0;
    return MMM_OK;
}

int mmm_p2p_enable(mmm_ctx* ctx, int on)
{
    if (!ctx) return MMM_ERR_ARG;
    MMM_CHECK(ctx, !on || (ctx->p2p && ctx->p2p->peer_h[ctx->rank]), "mmm_p2p_enable: not attached");
    ctx->p2p_on = on != 0;
    return MMM_OK;
}

/* "p2p" | "rccl" | "none" */
const char* mmm_comm_transport(const mmm_ctx* ctx)
{
    if (!ctx) return "none";
    if (ctx->p2p_on) return "p2p";
    return mmm_comm_active(ctx) ? "rccl" : "none";
}

} // extern "C"

// RCCL-bootstrapped set-up used by mmm_comm_init_rank: all-gather the handles, attach, rehearse, agree
int mmm_p2p_setup_over_rccl(mmm_ctx* ctx)
{
    // a one-rank communicator has nothing to exchange; MMM_P2P_ONE_RANK=1 sets the mailboxes up anyway, so that a single-GPU
    // box can run this function and the folded exchange end to end (tests/test_multirank_gpu.py)
    if (ctx->nranks > kP2PMaxRanks || (ctx->nranks < 2 && !getenv("MMM_P2P_ONE_RANK"))) return MMM_OK;
    if (const char* s = getenv("MMM_P2P")) if (atoi(s) == 0) return MMM_OK;
    const int n = ctx->nranks;
    // Every rank goes through the SAME sequence of collectives (all-gather, min, min) whatever fails locally: a local failure
    // only lowers ok_local.  The buffers of all three exist before the first collective, so an allocation failure cannot
    // leave the peers inside RCCL alone -- it is the one error returned before anything communicates.
    DevBuf<char> send, recv;
    DevBuf<double> flag;
    MMM_HIP(ctx, send.alloc(MMM_P2P_HANDLE_BYTES)); MMM_HIP(ctx, recv.alloc((size_t)n * MMM_P2P_HANDLE_BYTES)); MMM_HIP(ctx, flag.alloc(1));
    char mine[MMM_P2P_HANDLE_BYTES] = {0};
    int ok_local = 1;
    if (mmm_p2p_local_handle(ctx, n, mine) != MMM_OK) { ok_local = 0; memset(mine, 0, sizeof mine); }
    std::vector<char> all((size_t)n * MMM_P2P_HANDLE_BYTES, 0);
    if (hipMemcpyAsync(send.p, mine, MMM_P2P_HANDLE_BYTES, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) ok_local = 0;
    MMM_NCCL(ctx, ncclAllGather(send.p, recv.p, MMM_P2P_HANDLE_BYTES, ncclChar, ctx->comm, ctx->stream));
    if (hipMemcpyAsync(all.data(), recv.p, all.size(), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) ok_local = 0;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) ok_local = 0;
    if (ok_local && mmm_p2p_attach(ctx, n, ctx->rank, all.data()) != MMM_OK) ok_local = 0;
    // unanimous decisions: the copies around the min-reduction may fail locally (-> vote 0 / read 0), the collective itself is
    // always issued; an RCCL error is fatal for the communicator anyway and is the only early return
    auto agree = [&](int v, int* out) -> int {
        double d = v;
        if (hipMemcpyAsync(flag.p, &d, sizeof d, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) (void)hipMemsetAsync(flag.p, 0, sizeof d, ctx->stream);
        MMM_NCCL(ctx, ncclAllReduce(flag.p, flag.p, 1, ncclDouble, ncclMin, ctx->comm, ctx->stream));
        d = 0.0;
        if (hipMemcpyAsync(&d, flag.p, sizeof d, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) d = 0.0;
        *out = d > 0.5 ? 1 : 0;
        return MMM_OK;
    };
    int all_ok = 0, rc;
    // no. 1: everybody attached?  (a rank that could not must not leave the others polling)
    if ((rc = agree(ok_local, &all_ok))) return rc;
    if (!all_ok) { mmm_p2p_release(ctx); ctx->err.clear(); return MMM_OK; }
    // no. 2: everybody passed the rehearsal?
    bool good = false;
    if (p2p_selftest(ctx, &good) != MMM_OK) good = false;
    if ((rc = agree(good ? 1 : 0, &all_ok))) return rc;
    if (!all_ok) { ctx->p2p_on = false; ctx->err.clear(); }
    return MMM_OK;
}
