// ctx.hip -- context, stream and RCCL communicator management of libmmmusig_hip.so
#include "mmm_internal.h"

thread_local std::string g_mmm_create_error;

extern "C" {

int mmm_version(void) { return MMM_VERSION; }

int mmm_ctx_create(int device_id, mmm_ctx** out)
{
    if (!out) return mmm_fail(nullptr, MMM_ERR_ARG, "mmm_ctx_create: out == NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return mmm_fail(nullptr, MMM_ERR_NO_DEVICE, "mmm_ctx_create: no HIP device visible (%s)",
                        e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev)
        return mmm_fail(nullptr, MMM_ERR_ARG, "mmm_ctx_create: device %d out of range (0..%d)", device_id, ndev - 1);
    mmm_ctx* ctx = new mmm_ctx();
    ctx->device = device_id;
    hipDeviceProp_t prop;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
        int rc = mmm_fail(nullptr, MMM_ERR_HIP, "mmm_ctx_create: %s", hipGetErrorString(e));
        delete ctx;
        return rc;
    }
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    snprintf(ctx->arch, sizeof ctx->arch, "%s", prop.gcnArchName);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        int rc = mmm_fail(nullptr, MMM_ERR_NO_DEVICE, "mmm_ctx_create: device %d is %s; this library holds gfx950 code only",
                          device_id, prop.gcnArchName);
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return rc;
    }
    *out = ctx;
    return MMM_OK;
}

// Lifetime: ctx->refs counts the live models plus one reference held by the owner of the context (dropped by mmm_ctx_destroy).  Whoever
// drops the last reference runs the teardown, and nobody reads ctx after dropping a reference -- finalizers of a garbage-collected host run
// in any order and on any thread.
static void ctx_teardown(mmm_ctx* ctx)
{
    (void)hipSetDevice(ctx->device);
    // nothing in flight may still touch the events or the pinned block freed below -- whoever the caller is (the last model's destroy has
    // synchronised already; this does not depend on it)
    if (ctx->side) (void)hipStreamSynchronize(ctx->side);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (hipEvent_t e : ctx->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->pin_ev) if (e) (void)hipEventDestroy(e);
    if (ctx->pin_ctl) (void)hipHostFree(ctx->pin_ctl);
    mmm_p2p_release(ctx);
    if (ctx->comm) (void)ncclCommDestroy(ctx->comm);
    if (ctx->side) { (void)hipStreamSynchronize(ctx->side); (void)hipStreamDestroy(ctx->side); }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    delete ctx;
}

int mmm_ctx_destroy(mmm_ctx* ctx)
{
    if (!ctx) return MMM_OK;
    if (ctx->closing.exchange(true)) return MMM_DEFERRED;          // a second destroy while models are still alive: nothing left to give up
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);      // nothing in flight may still touch the pinned block, the events or a peer
    // give up everything that involves other ranks NOW, while they are still there.  Models that outlive their context (a
    // garbage-collected host) may only be destroyed after this call; a model of a multi-rank context that is USED afterwards gets an
    // error from its next call (its sums would silently be rank-local), see mmm_ctx_usable
    if (ctx->nranks > 1) ctx->closed_multi.store(true);
    mmm_p2p_release(ctx);
    if (ctx->comm) { (void)ncclCommDestroy(ctx->comm); ctx->comm = nullptr; }
    ctx->nranks = 1; ctx->rank = 0;
    if (ctx->refs.fetch_sub(1) - 1 > 0) return MMM_DEFERRED;        // ctx may be gone from here on
    ctx_teardown(ctx);
    return MMM_OK;
}

const char* mmm_last_error(const mmm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_mmm_create_error.c_str(); }

int mmm_ctx_synchronize(mmm_ctx* ctx)
{
    if (!ctx) return MMM_ERR_ARG;
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return mmm_p2p_check(ctx);
}

void* mmm_ctx_stream(mmm_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int mmm_ctx_device_name(mmm_ctx* ctx, char* buf, size_t n)
{
    if (!ctx || !buf || n == 0) return MMM_ERR_ARG;
    snprintf(buf, n, "%s", ctx->arch);
    return MMM_OK;
}

int mmm_ctx_profile_begin(mmm_ctx* ctx)
{
    if (!ctx) return MMM_ERR_ARG;
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->ev_used = 0;
    ctx->profiling = true;
    return MMM_OK;
}

int mmm_ctx_profile_repeat(mmm_ctx* ctx, int repeat)
{
    if (!ctx) return MMM_ERR_ARG;
    MMM_CHECK(ctx, repeat >= 1 && repeat <= 4, "mmm_ctx_profile_repeat: repeat must be 1..4");
    ctx->prof_repeat = repeat;
    return MMM_OK;
}

int mmm_ctx_profile_select(mmm_ctx* ctx, int phase)
{
    if (!ctx) return MMM_ERR_ARG;
    MMM_CHECK(ctx, phase >= 0 && phase <= 8, "mmm_ctx_profile_select: phase must be 0..7, or 8 for all of them");
    ctx->prof_phase = phase;
    return MMM_OK;
}

int mmm_ctx_profile_end_phases(mmm_ctx* ctx, int n_spans[8], double total_ms[8])
{
    if (!ctx || !n_spans || !total_ms) return MMM_ERR_ARG;
    ctx->profiling = false;
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 8; ++i) { n_spans[i] = 0; total_ms[i] = 0.0; }
    for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
        float ms = 0.f;
        MMM_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]));
        const int ph = ctx->ev_phase.size() > i / 2 ? (ctx->ev_phase[i / 2] & 7) : 0;
        n_spans[ph]++; total_ms[ph] += ms;
    }
    ctx->ev_used = 0;
    return MMM_OK;
}

int mmm_ctx_profile_end(mmm_ctx* ctx, int* n_launches, double* total_ms)
{
    if (!ctx || !n_launches || !total_ms) return MMM_ERR_ARG;
    ctx->profiling = false;
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double tot = 0.0;
    for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
        float ms = 0.f;
        MMM_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]));
        tot += ms;
    }
    *n_launches = (int)(ctx->ev_used / 2);
    *total_ms = tot;
    ctx->ev_used = 0;
    return MMM_OK;
}

int mmm_comm_unique_id(char out[MMM_UNIQUE_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) <= MMM_UNIQUE_ID_BYTES, "ncclUniqueId larger than the ABI slot");
    if (!out) return MMM_ERR_ARG;
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return mmm_fail(nullptr, MMM_ERR_RCCL, "ncclGetUniqueId: %s", ncclGetErrorString(r));
    memset(out, 0, MMM_UNIQUE_ID_BYTES);
    memcpy(out, &id, sizeof id);
    return MMM_OK;
}

int mmm_comm_init_rank(mmm_ctx* ctx, int nranks, int rank, const char id_bytes[MMM_UNIQUE_ID_BYTES])
{
    if (!ctx) return MMM_ERR_ARG;
    MMM_CHECK(ctx, nranks >= 1 && rank >= 0 && rank < nranks && id_bytes, "mmm_comm_init_rank: bad nranks/rank");
    MMM_CHECK(ctx, ctx->comm == nullptr, "mmm_comm_init_rank: communicator already initialised");
    MMM_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    MMM_NCCL(ctx, ncclCommInitRank(&ctx->comm, nranks, id, rank));
    ctx->nranks = nranks; ctx->rank = rank;
    // the per-iteration all-reduce goes over the xGMI mailboxes when they can be set up and rehearse correctly on every rank
    return mmm_p2p_setup_over_rccl(ctx);
}

int mmm_comm_nranks(const mmm_ctx* ctx) { return ctx ? ctx->nranks : 0; }

void mmm_tuning_opts_default(mmm_tuning_opts* o)
{
    if (o) memset(o, 0, sizeof *o);
}

int mmm_ctx_set_tuning(mmm_ctx* ctx, const mmm_tuning_opts* opts)
{
    if (!ctx) return MMM_ERR_ARG;
    mmm_tuning_opts t; mmm_tuning_opts_default(&t);
    if (opts) t = *opts;
    MMM_CHECK(ctx, t.lda_build >= MMM_BUILD_AUTO && t.lda_build <= MMM_BUILD_WIDE && t.ctm_build >= MMM_BUILD_AUTO && t.ctm_build <= MMM_BUILD_WIDE,
              "mmm_ctx_set_tuning: unknown build (lda_build %d, ctm_build %d)", t.lda_build, t.ctm_build);
    MMM_CHECK(ctx, t.geometry_cus >= 0 && t.grid_blocks >= 0 && t.waves_per_block >= 0 && t.moment_blocks >= 0 && t.resident_cap >= 0 && t.side_stream >= -1 && t.side_stream <= 1,
              "mmm_ctx_set_tuning: negative size or side_stream outside -1..1");
    // a caller built against a newer header must not have its choices dropped in silence
    MMM_CHECK(ctx, (t.disable & ~(unsigned)MMM_OFF_ALL) == 0, "mmm_ctx_set_tuning: unknown bits 0x%x in `disable` (this build knows 0x%x)", t.disable & ~(unsigned)MMM_OFF_ALL,
              (unsigned)MMM_OFF_ALL);
    for (int r : t.reserved) MMM_CHECK(ctx, r == 0, "mmm_ctx_set_tuning: a reserved field is %d, not 0 (a newer header's option?)", r);
    MMM_CHECK(ctx, t.solve_lanes == 0 || t.solve_lanes == 2 || t.solve_lanes == 8 || t.solve_lanes == 16 || t.solve_lanes == 32,
              "mmm_ctx_set_tuning: solve_lanes %d (0, 2, 8, 16 or 32)", t.solve_lanes);
    MMM_CHECK(ctx, t.solve_waves >= 0 && t.solve_waves <= 8, "mmm_ctx_set_tuning: solve_waves %d (0..8)", t.solve_waves);
    ctx->tune = t;
    return MMM_OK;
}

int mmm_ctx_get_tuning(const mmm_ctx* ctx, mmm_tuning_opts* out)
{
    if (!ctx || !out) return MMM_ERR_ARG;
    *out = ctx->tune;
    return MMM_OK;
}

void mmm_solver_opts_default(mmm_solver_opts* o)
{
    if (!o) return;
    o->xtol_rel = 1e-4; o->xtol_abs = 1e-4; o->nu_lower = 1e-7; o->xtol_rule = 0; o->max_eval = 2000;
}

} // extern "C"

void mmm_ctx_model_created(mmm_ctx* ctx) { ctx->refs.fetch_add(1); }
void mmm_ctx_model_destroyed(mmm_ctx* ctx)
{
    if (ctx->refs.fetch_sub(1) - 1 == 0) ctx_teardown(ctx);      // the owner's reference went first (MMM_DEFERRED) and this was the last model
}

int mmm_ctx_usable(mmm_ctx* ctx, const char* what)
{
    if (ctx->closed_multi.load())
        return mmm_fail(ctx, MMM_ERR_ARG, "%s: the context of this model was destroyed while it had %s; the model can only be destroyed", what,
                        "a communicator (its sums would be rank-local)");
    return MMM_OK;
}
