/*
 * mmmusig.h -- C ABI of libmmmusig_hip.so: the MI355X (gfx950) variational-EM backend that sits under the
 * MultiModalMuSig.jl API (LDA / ILDA / MMCTM / IMMCTM: fit!, the update_*! functions, transform, fit_heldout,
 * predict_modality_η, and the restart sweep of scripts/run_mmctm.jl as batched fits).
 *
 * The reference has no FFI: its hot path is reached by ordinary Julia dispatch.  Each entry point below
 * therefore names the Julia function (file:line under the reference's src/) whose work it replaces; the Julia
 * shim that `ccall`s them is multimodalmusig.jl_amd/julia/MultiModalMuSigHIP.jl.
 *
 * Conventions
 *   - plain C types only; every function returns an int status (0 = MMM_OK, negative = error);
 *     nothing throws across the ABI; mmm_last_error() returns the message of the last failure.
 *   - all reals are double; term ids are 0-based int32, counts int32, CSR offsets int64 (the shim converts
 *     the reference's 1-based Int64 `X[d][:,1]`).
 *   - host pointers in/out unless a name ends in `_dev`; the library copies inside the call and keeps no
 *     host pointer after return.  Model state lives in HBM between calls, owned by the handle.
 *   - one mmm_ctx per host thread / process = one GPU + one HIP stream (+ one RCCL rank when
 *     mmm_comm_init_rank has been called).  Calls on a ctx are serialised by the caller.
 *   - array layouts are exactly the reference's Julia (column-major) layouts flattened:
 *       LDA    lambda/Elnbeta/beta  V x K   [v + V*k]        (LDA.jl:8-10)
 *              gamma/Elntheta/theta K x D   [k + K*d]        (LDA.jl:13-15)
 *              phi   per doc K x W_d, docs concatenated      [K*doc_ptr[d] + k + K*w]   (LDA.jl:16)
 *       MMCTM  doc_ptr M*(D+1) absolute offsets into the modality-major concatenated term/count arrays
 *              lambda/nu/props MK x D [i + MK*d]; zeta M x D [m + M*d]
 *              gamma/Elnphi/phi [goff[m] + k*V[m] + v]
 *              theta [toff[m] + (e - doc_ptr[m*(D+1)])*K[m] + k]   (e = absolute entry index)
 *              mu MK; Sigma/invSigma MK x MK column-major
 *       IMMCTM features [foff[m] + i*V[m] + v] 0-based; gamma/Elnphi [goff[m] + k*SJ[m] + joff[m][i] + j]
 */
#ifndef MMMUSIG_H
#define MMMUSIG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMM_VERSION 123

enum {
    MMM_OK = 0,
    MMM_DEFERRED = 1,         /* mmm_ctx_destroy with live models: communicator and mailboxes released now, the rest with the last model */
    MMM_ERR_ARG = -1,         /* bad argument / inconsistent sizes                         */
    MMM_ERR_HIP = -2,         /* a HIP runtime call failed (message has hipGetErrorString) */
    MMM_ERR_RCCL = -3,        /* an RCCL call failed                                       */
    MMM_ERR_UNSUPPORTED = -4, /* shape outside what the kernels are built for              */
    MMM_ERR_NUMERIC = -5,     /* singular Sigma in the M-step                              */
    MMM_ERR_NO_DEVICE = -6    /* no gfx950 device visible                                  */
};

typedef struct mmm_ctx mmm_ctx;
typedef struct mmm_lda mmm_lda;
typedef struct mmm_ctm mmm_ctm;

/* ---- context ------------------------------------------------------------------------------------------ */
int mmm_version(void);
/* Create a context on HIP device `device_id` with its own non-blocking stream. */
int mmm_ctx_create(int device_id, mmm_ctx** out);
/* With models still alive on it (a garbage-collected host destroys in no particular order) the context gives up its communicator and
 * mailboxes at once and returns MMM_DEFERRED; stream and memory go with the last mmm_*_destroy.  Such models may be destroyed; if the
 * context had a communicator (nranks > 1) every other call on them returns MMM_ERR_ARG -- their sums would silently be rank-local. */
int mmm_ctx_destroy(mmm_ctx* ctx);
/* Message of the last error on this ctx (ctx == NULL: last error of a failed mmm_ctx_create). */
const char* mmm_last_error(const mmm_ctx* ctx);
int mmm_ctx_synchronize(mmm_ctx* ctx);
/* The hipStream_t every kernel of this ctx is launched on (for event timing by the caller). */
void* mmm_ctx_stream(mmm_ctx* ctx);
int mmm_ctx_device_name(mmm_ctx* ctx, char* buf, size_t n);

/* ---- run-time choices of the caller (SURVEY section 5 "Config / flags": a plain C struct across the ABI) -------------------------------
 * What a caller may legitimately choose -- which E-step build a handle takes, a PINNED launch geometry, the side stream -- travels here,
 * not in environment variables: mmm_ctx_set_tuning stores the options on the context and every handle CREATED afterwards keeps its own
 * copy (two handles of one process may differ).  0 in a field = the library decides.  The all-reduce transport is chosen separately
 * (mmm_p2p_enable).  The only environment variables the library still reads concern the transport set-up: MMM_P2P, MMM_P2P_TIMEOUT_S,
 * MMM_P2P_ONE_RANK, MMM_FORCE_RCCL. */
enum { MMM_BUILD_AUTO = 0,
       MMM_BUILD_SPARSE = 1,   /* sweep the CSR arrays / padded rows through the LDS tables (any corpus)                              */
       MMM_BUILD_DENSE = 2,    /* rows of counts, statistics in registers (dense corpora; falls back when the shape has no such build) */
       MMM_BUILD_WIDE = 3 };   /* topic tables through L2 instead of LDS, term-major statistics sweep (large vocabularies)             */
enum { /* mmm_tuning_opts.disable: optimisations a test or an A/B run may switch off, one bit each (results are unchanged unless noted) */
    MMM_OFF_LDA_PADDED_ROWS = 1 << 0,   /* single-step E-step build: fetch (term, count) through doc_ptr instead of padded rows                  */
    MMM_OFF_LDA_COUNT_ROWS = 1 << 1,    /* dense corpora: keep (term, count) rows instead of rows of counts                                        */
    MMM_OFF_LDA_ROWS16 = 1 << 2,        /* rows of int32 instead of 16-bit counts                                                                  */
    MMM_OFF_LDA_LL_JOIN = 1 << 3,       /* reduce blocks do not join the log-likelihood sweep of the merged launch                                 */
    MMM_OFF_LDA_MERGED = 1 << 4,        /* reduce, ll and M-step as separate launches instead of k_lda_reduce_ll_mstep                             */
    MMM_OFF_P2P_FOLDED = 1 << 5,        /* several GPUs: the mailbox exchange as its own launch instead of folded into the reduce / M-step blocks */
    MMM_OFF_CTM_PACKED = 1 << 6,        /* solve phase: no packed document groups (sum K = 6 / 10 / 12 lanes per document)                         */
    MMM_OFF_CTM_CPL = 1 << 7,           /* solve phase: one coordinate per lane everywhere (no several-coordinates-per-lane builds)                */
    MMM_OFF_CTM_KFIT = 1 << 8,          /* theta phase: 16-wide topic loops for every shape                                                        */
    MMM_OFF_CTM_FUSED_GAUSS = 1 << 9,   /* Gaussian M-step as its own launch instead of block 0 of the log-likelihood launch                       */
    MMM_OFF_CTM_LL_ROWS = 1 << 10,      /* handles with rows of counts: props / log-likelihood sweep over the CSR arrays                           */
    MMM_OFF_LDA_EARLY_PROLOGUE = 1 << 11, /* single-step E-step build: every pass forms its own Elntheta / exp(Elntheta) instead of the previous pass's merged launch */
    MMM_OFF_CTM_PIPE_GAUSS = 1 << 12,    /* Gaussian M-step: the three-barrier-per-column inversion instead of the pipelined one (sum K <= 32); same bits         */
    MMM_OFF_CTM_SOLVE_ORDER = 1 << 13,   /* solve phase: a wave's slots take its documents in index order instead of longest-lambda-solve-of-the-previous-pass first; same bits */
    MMM_OFF_ALL = (1 << 14) - 1          /* every bit this build knows; mmm_ctx_set_tuning rejects others (and non-zero reserved fields) with MMM_ERR_ARG */
};
typedef struct {
    int lda_build;        /* MMM_BUILD_*: E-step build of LDA / ILDA handles                                                              */
    int ctm_build;        /* MMM_BUILD_*: theta phase of the fused pass of MMCTM / IMMCTM handles                                         */
    int geometry_cus;     /* > 0: size every launch whose block count fixes the association of a cross-document sum (and so the bits of  */
                          /* a fit) as if the device had this many CUs -- the same value gives the same bits on any gfx950 device or     */
                          /* partition mode (256 = an MI355X in SPX mode).  0: the device's own CU count.                                */
    int grid_blocks;      /* > 0: that many blocks for the E-step kernel (LDA: forces the grid-stride build) / the theta phase (CTM)      */
    int waves_per_block;  /* > 0: waves per block of the LDA E-step kernel                                                                */
    int moment_blocks;    /* > 0: blocks of the CTM moment sums                                                                           */
    int side_stream;      /* CTM fit passes on one GPU: -1 never, 0 library's choice (IMMCTM only), 1 whenever possible                   */
    int resident_cap;     /* > 0: lowers the residency bound of the merged LDA launch (tests of the fallback)                             */
    unsigned disable;     /* MMM_OFF_* bits                                                                                               */
    int solve_lanes;      /* CTM solve phase: lanes per document, 0 = the library's choice by shape and corpus size (mmm_ctm_geometry out[5]); */
                          /* sum K = 10: 2 (x 5 coordinates) or 8 (x 2); sum K = 28: 16 (x 2) or 32 (x 1).  The layouts associate a document's  */
                          /* sums differently (sum K = 10) -- the value pins the bits across corpus sizes                                   */
    int solve_waves;      /* > 0: persistent solve kernels run with this many waves per SIMD (1..8) instead of the library's choice           */
    int reserved[5];      /* 0                                                                                                            */
} mmm_tuning_opts;
void mmm_tuning_opts_default(mmm_tuning_opts* o);
/* opts == NULL: back to the defaults.  Applies to handles created on ctx from now on. */
int mmm_ctx_set_tuning(mmm_ctx* ctx, const mmm_tuning_opts* opts);
int mmm_ctx_get_tuning(const mmm_ctx* ctx, mmm_tuning_opts* out);
/* HIP-event timing of the dominant kernel of the hot path (the fused E-step kernel) on the ctx stream: between
 * begin and end every launch of it is bracketed by an event pair; end synchronises and returns the number of
 * launches and the sum of their durations in milliseconds. */
int mmm_ctx_profile_begin(mmm_ctx* ctx);
/* repeat = 2: every profiled span holds the (idempotent) LDA E-step kernel twice; the difference of the spans measured with
 * repeat 2 and repeat 1 is the kernel's duration without the ~4 us that an event pair adds around a single launch. */
int mmm_ctx_profile_repeat(mmm_ctx* ctx, int repeat);
/* which launches the spans bracket: 0 (default) the dominant kernel; LDA 1 = everything of a pass after the E-step kernel
 * (reduction, ll sweep, M-step); CTM 1 = theta phase, 2 = moments + reduction + M-step, 3 = props / log-likelihood launches;
 * Lets bench.py account for the whole iteration kernel by kernel (its `iteration` block). */
int mmm_ctx_profile_select(mmm_ctx* ctx, int phase);
/* phase 8 brackets every phase at once; mmm_ctx_profile_end_phases then returns span counts and summed durations per phase (0..7).  For
 * passes of milliseconds (CTM), where the few microseconds an event pair adds do not matter. */
int mmm_ctx_profile_end_phases(mmm_ctx* ctx, int n_spans[8], double total_ms[8]);
int mmm_ctx_profile_end(mmm_ctx* ctx, int* n_launches, double* total_ms);

/* ---- multi-GPU: documents are sharded across ranks, sufficient statistics are all-reduced (RCCL) ---------
 * (no counterpart in the reference, which is single-threaded; replaces the doc loops' cross-document sums
 *  LDA.jl:103-105, MMCTM.jl:201,205-210,230-240 and the log-likelihood sums).
 * With a communicator on ctx, every model call that produces or consumes a cross-document sum is COLLECTIVE: all ranks make
 * the same calls on their handles in the same order -- *_create, *_iterate, *_fit, *_infer, the update_* stages, *_loglik,
 * *_elbo, and *_ll_history / *_set (they complete the last pass's pending log-likelihood first).  *_get and *_destroy are
 * local. */
#define MMM_UNIQUE_ID_BYTES 128
int mmm_comm_unique_id(char out[MMM_UNIQUE_ID_BYTES]);                 /* rank 0: ncclGetUniqueId   */
int mmm_comm_init_rank(mmm_ctx* ctx, int nranks, int rank, const char id[MMM_UNIQUE_ID_BYTES]);
int mmm_comm_nranks(const mmm_ctx* ctx);
/* The all-reduce transport.  The payload is 1-20 KB once per iteration, so latency decides: after mmm_comm_init_rank the
 * library all-reduces through per-rank MAILBOXES in fine-grained device memory that the peers write directly over xGMI
 * (one kernel per call, sums in rank order: same bits on every rank), provided the IPC mappings could be made and a
 * known-answer rehearsal passed on every rank; otherwise, and for payloads above the mailbox capacity, ncclAllReduce.
 * MMM_P2P=0 in the environment keeps RCCL.  mmm_comm_transport: "p2p", "rccl" or "none".
 * The mailbox path can also be set up without RCCL: every rank calls mmm_p2p_local_handle, the host exchanges the
 * handles (nranks x MMM_P2P_HANDLE_BYTES, rank-major), every rank calls mmm_p2p_attach and mmm_p2p_selftest, and all
 * ranks call mmm_p2p_enable(ctx, 0) unless every rank's rehearsal passed. */
#define MMM_P2P_HANDLE_BYTES 64
const char* mmm_comm_transport(const mmm_ctx* ctx);
int mmm_p2p_local_handle(mmm_ctx* ctx, int nranks, char out[MMM_P2P_HANDLE_BYTES]);
int mmm_p2p_attach(mmm_ctx* ctx, int nranks, int rank, const char* handles);
int mmm_p2p_selftest(mmm_ctx* ctx, int* ok);
int mmm_p2p_enable(mmm_ctx* ctx, int on);

/* ---- LDA (src/LDA.jl) ----------------------------------------------------------------------------------- */
enum { /* field ids for mmm_lda_get / mmm_lda_set; sizes in doubles */
    MMM_LDA_LAMBDA = 0,   /* V*K  */
    MMM_LDA_ELNBETA = 1,  /* V*K  */
    MMM_LDA_BETA = 2,     /* V*K  */
    MMM_LDA_GAMMA = 3,    /* K*D  */
    MMM_LDA_ELNTHETA = 4, /* K*D  */
    MMM_LDA_THETA = 5,    /* K*D  */
    MMM_LDA_PHI = 6,      /* K*nnz */
    /* ILDA handles (mmm_ilda_create): the factor arrays of ILDA.jl:6-9, lambda[i] J_i x K column-major at K*sum_{q<i} J_q;
     * for such handles ELNBETA / BETA above are the effective V x K tables sum_i Elnβ[i][f_vi,k] / prod_i β[i][f_vi,k] */
    MMM_ILDA_LAMBDA = 7, MMM_ILDA_ELNBETA = 8, MMM_ILDA_BETA = 9      /* sum(J)*K */
};
/* Constructor LDA(k, alpha, eta, V, X) -- LDA.jl:24-54.  lambda0 (V*K) is the random init the shim draws with
 * rand(1:100, V, K) (LDA.jl:36); the ctor state gamma=1, phi=1/K, Elnbeta, Elntheta is built on the GPU.
 * With an RCCL communicator on ctx, (D, doc_ptr, term, count) is THIS RANK'S shard of the documents.
 * Shapes: any V; K <= 256 (tuned builds to 32 topics, rolled-loop kernels beyond); MMM_ERR_UNSUPPORTED otherwise -- the reference has no limit. */
int mmm_lda_create(mmm_ctx* ctx, int D, int V, int K, double alpha, double eta, const int64_t* doc_ptr,
                   const int32_t* term, const int32_t* count, const double* lambda0, mmm_lda** out);
/* Constructor ILDA(k, alpha, eta::Vector, features, X) -- ILDA.jl:25-56: the topic-term distribution factorises over the I
 * features of a term (features: [i*V + v], 0-based values < J[i]).  lambda0: rand(1:100, J_i, K) per feature (ILDA.jl:36).
 * The handle is an mmm_lda: every mmm_lda_* entry point works on it (update_ϕ!/γ!/λ!/β! ILDA.jl:65-130, ll :203-239,
 * ELBO :132-201 -- including the reference's ElnQβ, which keeps only the last feature's term, :175-182 --, fit! :246-272,
 * and the frozen-topic passes of fit_heldout :320-353). */
int mmm_ilda_create(mmm_ctx* ctx, int D, int V, int K, double alpha, int I, const int* J, const double* eta,
                    const int32_t* features, const int64_t* doc_ptr, const int32_t* term, const int32_t* count,
                    const double* lambda0, mmm_lda** out);
int mmm_lda_destroy(mmm_lda* m);
int mmm_lda_get(mmm_lda* m, int field, double* host, size_t n);
int mmm_lda_set(mmm_lda* m, int field, const double* host, size_t n);
/* The hyper-parameters are plain mutable fields upstream (`model.α = 0.5; fit!(model)`, LDA.jl:2-12 / ILDA.jl:2-12): alpha, and eta as
 * n_eta = 1 value (LDA) or one per feature (ILDA).  Takes effect from the next update_γ! / update_λ! on. */
int mmm_lda_set_hyper(mmm_lda* m, double alpha, const double* eta, int n_eta);
/* One-to-one GPU counterparts of the reference's update functions (stage API; used by the parity tests) */
int mmm_lda_update_gamma(mmm_lda* m);   /* update_γ!  LDA.jl:82-90  (+ update_Elnθ! :78-80)  */
int mmm_lda_update_phi(mmm_lda* m);     /* update_ϕ!  LDA.jl:69-76                           */
int mmm_lda_update_lambda(mmm_lda* m);  /* update_λ!  LDA.jl:100-108 (+ update_Elnβ! :96-98) */
int mmm_lda_update_beta(mmm_lda* m);    /* update_β!  LDA.jl:110-112                         */
int mmm_lda_update_Elntheta(mmm_lda* m);/* update_Elnθ! alone  LDA.jl:78-80   (the reference calls it at the end of update_γ!) */
int mmm_lda_update_Elnbeta(mmm_lda* m); /* update_Elnβ! alone  LDA.jl:96-98 / ILDA.jl:96-101 (... at the end of update_λ!)    */
int mmm_lda_update_theta(mmm_lda* m);   /* update_θ!  LDA.jl:92-94                           */
int mmm_lda_loglik(mmm_lda* m, double* ll);                    /* calculate_loglikelihood LDA.jl:174-196 */
int mmm_lda_elbo(mmm_lda* m, double* elbo, double terms[7]);   /* calculate_elbo LDA.jl:114-172          */
/* Fused hot path: n_iter passes of { update_γ!; update_ϕ!; update_λ!; update_β!; update_θ!; ll } (the body of
 * fit!, LDA.jl:201-209) enqueued on the ctx stream without host synchronisation.  ll of pass i is written
 * to an internal device history; read it with mmm_lda_ll_history. */
int mmm_lda_iterate(mmm_lda* m, int n_iter);
int mmm_lda_ll_history(mmm_lda* m, double* ll, int max_n, int* n);
/* Which E-step build the handle uses and its launch geometry (diagnostics, tests): out[0] = lanes per document, [1] = blocks,
 * [2] = waves per block, [3] = 1 for the single-step build (small corpora: the grid covers every document at once), [4] = 1 for
 * the wide-table path (tables beyond LDS), [5] = 1 for the dense-row build (dense corpus over <= 128 terms: rows of counts,
 * statistics accumulated in registers; 2: its 32-lane variant), [6] = its term slots per lane, [7] = topics padded to. */
int mmm_lda_geometry(const mmm_lda* m, int out[8]);
/* Bytes of corpus the E-step build reads per document when the handle keeps rows (rows of 16- or 32-bit counts: 2 or 4 bytes x 16 x
 * slots per lane; padded (term,count) rows: 8 x V); 0 when it sweeps the CSR arrays (8 bytes per nonzero + offsets).  bench.py's
 * "algorithmic bytes as implemented". */
int mmm_lda_row_bytes(const mmm_lda* m);
/* 1 once fused passes of this handle have formed the NEXT pass's Elntheta / exp(Elntheta) inside their merged launch (single-step build: the
 * E-step kernel of every pass but the first of a call then reads exp(Elntheta) and writes gamma only -- bench.py counts its bytes
 * accordingly), else 0. */
int mmm_lda_prologue_moved(const mmm_lda* m);
/* fit!(model; maxiter, tol) -- LDA.jl:198-224: iterate until |dll|/|ll| < tol after > 10 passes, then ELBO. */
int mmm_lda_fit(mmm_lda* m, int maxiter, double tol, double* ll_hist, int* n_iter, int* converged,
                double* elbo);
/* Frozen-topic inference on a model whose topics were uploaded with mmm_lda_set: the loop of
 * transform(model, X) LDA.jl:233-263 (unsmoothed = 1: unsmoothed_update_ϕ!, phi ∝ exp(Elnθ)·β, needs BETA) or of
 * fit_heldout(Xheldout, model) LDA.jl:265-295 (unsmoothed = 0: update_ϕ! with Elnβ, needs ELNBETA and BETA):
 * { update_γ!; (unsmoothed_)update_ϕ!; update_θ!; ll } until the stopping rule fires after > 10 passes.  Topics are
 * not touched; γ, Elnθ, θ, ϕ of the handle's documents are the result. */
int mmm_lda_infer(mmm_lda* m, int unsmoothed, int maxiter, double tol, double* ll_hist, int* n_iter,
                  int* converged);

/* ---- MMCTM (src/MMCTM.jl) and IMMCTM (src/IMMCTM.jl) -------------------------------------------------------- */
typedef struct {
    double xtol_rel;     /* 1e-4   MMCTM.jl:129,158 */
    double xtol_abs;     /* 1e-4   MMCTM.jl:130,159 */
    double nu_lower;     /* 1e-7   MMCTM.jl:157     */
    int xtol_rule;       /* 0: NLopt >= 2.7 stopping rule, 1: NLopt <= 2.6 (Project.toml:15 allows both) */
    int max_eval;        /* safety cap on objective evaluations per MMA solve (NLopt: unlimited); 0 -> 2000 */
} mmm_solver_opts;
void mmm_solver_opts_default(mmm_solver_opts* o);

enum { /* field ids for mmm_ctm_get / mmm_ctm_set */
    MMM_CTM_MU = 0, MMM_CTM_SIGMA = 1, MMM_CTM_INVSIGMA = 2, MMM_CTM_GAMMA = 3, MMM_CTM_ELNPHI = 4,
    MMM_CTM_PHI = 5, MMM_CTM_LAMBDA = 6, MMM_CTM_NU = 7, MMM_CTM_ZETA = 8, MMM_CTM_PROPS = 9,
    MMM_CTM_THETA = 10, MMM_CTM_ALPHA = 11
};
/* Constructor MMCTM(k, alpha, V, X) -- MMCTM.jl:29-91 (init = :random; gamma0 is the rand(1:100, V[m]) draw per
 * topic, MMCTM.jl:60-63).  n_feat/J/features == NULL: MMCTM.  Otherwise IMMCTM(k, alpha, features, X) --
 * IMMCTM.jl:29-78 with alpha of length sum_m I[m] and gamma0 in the IMMCTM layout.
 * Shapes: any V[m]; M <= 8; K[m] <= 64; sum K <= 256 (tuned kernels to sum K = 64 with K[m] <= 32, generic ones beyond);
 * MMM_ERR_UNSUPPORTED otherwise -- the reference has no limit. */
int mmm_ctm_create(mmm_ctx* ctx, int D, int M, const int* K, const int* V, const double* alpha,
                   const int64_t* doc_ptr, const int32_t* term, const int32_t* count, const int* n_feat,
                   const int* J, const int32_t* features, const double* gamma0, const mmm_solver_opts* opts,
                   mmm_ctm** out);
int mmm_ctm_destroy(mmm_ctm* m);
int mmm_ctm_get(mmm_ctm* m, int field, double* host, size_t n);
int mmm_ctm_set(mmm_ctm* m, int field, const double* host, size_t n);
/* stage API (GPU counterparts of the reference's per-document and M-step functions) */
int mmm_ctm_update_zeta(mmm_ctm* m);    /* update_ζ! for every doc      MMCTM.jl:172-181              */
int mmm_ctm_update_theta(mmm_ctm* m);   /* update_θ! for every doc      MMCTM.jl:183-198 / IMMCTM.jl:152-172 */
int mmm_ctm_update_nu(mmm_ctm* m);      /* update_ν! for every doc      MMCTM.jl:156-170 (LD_MMA)     */
int mmm_ctm_update_lambda(mmm_ctm* m);  /* update_λ! for every doc      MMCTM.jl:127-143 (LD_MMA)     */
int mmm_ctm_update_mu(mmm_ctm* m);      /* update_μ!                    MMCTM.jl:200-202              */
int mmm_ctm_update_Sigma(mmm_ctm* m);   /* update_Σ!                    MMCTM.jl:204-212              */
int mmm_ctm_update_gamma(mmm_ctm* m);   /* update_γ! (+ update_Elnϕ!)   MMCTM.jl:224-242,214-222      */
int mmm_ctm_update_Elnphi(mmm_ctm* m);  /* update_Elnϕ!                 MMCTM.jl:214-222 / IMMCTM.jl:188-197 */
int mmm_ctm_update_alpha(mmm_ctm* m);   /* update_α! (1-D LD_MMA per α) MMCTM.jl:252-269, IMMCTM.jl:225-244 */
int mmm_ctm_update_props(mmm_ctm* m);   /* update_props!                MMCTM.jl:145-154              */
int mmm_ctm_update_phi(mmm_ctm* m);     /* update_ϕ!                    MMCTM.jl:244-250              */
/* The reference's per-document functions take the document: update_ζ!(model, d), update_θ!(model, d), update_ν!(model, d),
 * update_λ!(model, d) (MMCTM.jl:127-198; test/mmctm.jl:92-199 call them that way).  Only document d (0-based) changes. */
enum { MMM_STAGE_ZETA = 0, MMM_STAGE_THETA = 1, MMM_STAGE_NU = 2, MMM_STAGE_LAMBDA = 3 };
int mmm_ctm_update_doc(mmm_ctm* m, int stage, int d);
/* calculate_sumθ(model, d) and calculate_Ndivζ(model, d) -- MMCTM.jl:110-125 / IMMCTM.jl:90-105: sum K doubles each (either may be NULL),
 * from the stored θ and ζ of document d (0-based) */
int mmm_ctm_doc_sums(mmm_ctm* m, int d, double* sumtheta, double* Ndivzeta);
int mmm_ctm_loglik(mmm_ctm* m, double* ll /* M */);                /* calculate_loglikelihoods MMCTM.jl:384-448 */
int mmm_ctm_elbo(mmm_ctm* m, double* elbo, double terms[7]);       /* calculate_elbo MMCTM.jl:271-382           */
/* objective/gradient of one document evaluated by the device code the MMA solves use (common.jl:11-36) */
int mmm_ctm_objectives(mmm_ctm* m, int d, double* lambda_val, double* lambda_grad, double* nu_val,
                       double* nu_grad);
/* solver statistics of the last E-step: total objective evaluations of the nu and lambda solves, number of
 * solves that hit max_eval, and (optional, D ints each) per-document evaluation counts */
int mmm_ctm_solver_stats(mmm_ctm* m, int64_t* n_eval_nu, int64_t* n_eval_lambda, int64_t* n_capped,
                         int* per_doc_nu, int* per_doc_lambda);
/* Non-fatal events, COUNTED instead of raised (SURVEY section 8b "error convention": the reference ignores NLopt's return code, MMCTM.jl:141,
 * 168, never checks for NaN, and carries on -- so does the library): out[0] = LD_MMA solves of the last E-step that hit max_eval (= n_capped
 * above), out[1] = LD_MMA solves of the last E-step in which an objective value was not finite (a NaN / Inf in a document's λ, ν, ζ or in
 * μ / invΣ: such a solve cannot satisfy NLopt's tests and runs into the cap), out[2] = values of the handle's log-likelihood history that
 * are not finite (a fit whose ll is NaN never meets the stopping rule of common.jl:48-56 and runs to maxiter), out[3] = 0.  LDA / ILDA
 * handles have no solves: out[0] = out[1] = 0. */
int mmm_ctm_events(mmm_ctm* m, int64_t out[4]);
int mmm_lda_events(mmm_lda* m, int64_t out[4]);
/* Launch geometry that fixes the ORDER of the sums across documents (and so their bits): out[0] = lanes per document L,
 * [1] = theta-phase blocks, [2] = waves per theta-phase block, [3] = blocks of the moment sums, [4] = 1 when the handle
 * takes the wide-table path (term-major posting sweep instead of LDS slabs), 2 when the fused pass's theta phase runs over rows of counts
 * (dense corpora: 16 lanes per document, statistics in registers), [5] = lanes per document in the solve phase (L, sum K for the packed
 * builds, or 2 / 4), [6] = coordinates per lane in the solve phase, [7] = waves of the persistent solve launch (each takes a contiguous
 * range of D / [7] documents; 0: one document group per wave step -- no bits depend on it).  The parity tests hand it to the
 * order-matched CPU restatement (oracle/mmm_twin.c), which then reproduces a whole fit bit for bit. */
int mmm_ctm_geometry(const mmm_ctm* m, int out[8]);
/* Parity probe: out[i] = op(a[i], b[i]) evaluated by the device functions the kernels use (csrc/mmm_arith.h, dev_math.h).
 * op 0 exp, 1 log, 2 digamma (x > 0), 3 a / b, 4 sqrt, 5 / 6 / 7 sum over consecutive groups of 16 / 32 / 64 values in the
 * lane-butterfly order of the document groups (out[i] = total of i's group; n a multiple of 64), 8 the full-wave butterfly,
 * 9 / 10 the table-driven exp / log of the LD_MMA objectives (ar_exp_tab, ar_log_tab). */
int mmm_debug_math(mmm_ctx* ctx, int op, size_t n, const double* a, const double* b, double* out);
/* Fused hot path: n_iter passes of the body of fit! (MMCTM.jl:462-479 / IMMCTM.jl:440-451) */
/* fit_flags: keyword arguments of fit! (MMCTM.jl:457-458): MMM_FIT_UPDATE_SIGMA = updateΣ (IMMCTM always updates Σ,
 * IMMCTM.jl:445), MMM_FIT_AUTO_ALPHA = autoα (update_α! after update_γ!, MMCTM.jl:472-474) */
enum { MMM_FIT_UPDATE_SIGMA = 1, MMM_FIT_AUTO_ALPHA = 2 };
int mmm_ctm_iterate(mmm_ctm* m, int n_iter, int fit_flags);
int mmm_ctm_ll_history(mmm_ctm* m, double* ll /* M*max_n */, int max_n, int* n);
int mmm_ctm_fit(mmm_ctm* m, int maxiter, double tol, int fit_flags, double* ll_hist /* M*maxiter */,
                int* n_iter, int* converged, double* elbo);

/* Frozen-topic inference on a model whose globals were uploaded with mmm_ctm_set.  Every pass runs the document loop
 * { update_ζ!; update_θ!; update_ν!; update_λ! }, update_props! and the log-likelihoods; the stopping rule applies after
 * > 10 passes.  flags = 0: fit_heldout / predict_modality_η (MMCTM.jl:554-634, IMMCTM.jl:468-545; needs MU, SIGMA,
 * INVSIGMA, GAMMA, ELNPHI, PHI).  MMM_INFER_UNSMOOTHED: unsmoothed_update_θ! instead of update_θ! (θ ∝ exp(λ)·ϕ,
 * MMCTM.jl:496-509; needs PHI) -- with it, transform (MMCTM.jl:511-552).  MMM_INFER_FIT_GAUSSIAN: also update_μ! and
 * update_Σ! every pass (transform's fit_gaussian).  Topics are not touched. */
enum { MMM_INFER_UNSMOOTHED = 1, MMM_INFER_FIT_GAUSSIAN = 2 };
int mmm_ctm_infer(mmm_ctm* m, int flags, int maxiter, double tol, double* ll_hist /* M*maxiter */, int* n_iter,
                  int* converged);

/* Restart batching -- scripts/run_mmctm.jl:77-134 fits the same corpus from several random initialisations and
 * keeps the best.  A batch handle holds R independent models ("replicas") over ONE resident corpus; the batched
 * fit advances all of them with one launch per kernel (replica on grid.y), so the many small-corpus fits of a
 * restart sweep fill the GPU together.  Replica r computes exactly what a model created from gamma0[r] alone
 * computes.  gamma0: R contiguous blocks in the layout of mmm_ctm_create.
 * The per-model API above (get/set/update_* /loglik/elbo/iterate/fit) acts on the selected replica (0 after
 * create).  theta is kept for one replica at a time and rebuilt when the selection changes. */
int mmm_ctm_create_batch(mmm_ctx* ctx, int R, int D, int M, const int* K, const int* V, const double* alpha,
                         const int64_t* doc_ptr, const int32_t* term, const int32_t* count, const int* n_feat,
                         const int* J, const int32_t* features, const double* gamma0,
                         const mmm_solver_opts* opts, mmm_ctm** out);
int mmm_ctm_replicas(const mmm_ctm* m);
int mmm_ctm_select(mmm_ctm* m, int r);
/* fit! of every replica, in lock step; a replica stops when its own stopping rule fires (common.jl:48-51).
 * ll_hist: [R][maxiter][M]; n_iter, converged: [R]; elbo: [R] or NULL.  All replicas must have the same number
 * of earlier passes. */
int mmm_ctm_fit_batch(mmm_ctm* m, int maxiter, double tol, int fit_flags, double* ll_hist, int* n_iter,
                      int* converged, double* elbo);

/* ---- free functions of the reference (arguments are caller arrays, no model) --------------------------------------
 * The reference's tests call these directly (test/common.jl:79-97; test/mmctm.jl:135-148,268-293,349-388; test/immctm.jl:122-160,
 * 273-294,350-386).  Values and gradients in the reference's MAXIMISATION form; grad may be NULL (the reference's `length(∇) > 0`). */
/* λ_objective(λ, ∇λ, ν, Ndivζ, sumθ, μ, invΣ) -- common.jl:11-23; invSigma n x n column-major */
int mmm_lambda_objective(mmm_ctx* ctx, int n, const double* lambda, const double* nu, const double* Ndivzeta, const double* sumtheta,
                         const double* mu, const double* invSigma, double* val, double* grad);
/* ν_objective(ν, ∇ν, λ, Ndivζ, μ, invΣ) -- common.jl:25-36 (μ is unused there too and may be NULL) */
int mmm_nu_objective(mmm_ctx* ctx, int n, const double* nu, const double* lambda, const double* Ndivzeta, const double* mu,
                     const double* invSigma, double* val, double* grad);
/* α_objective(α, ∇α, sum_Elnϕ, K, V) -- common.jl:38-46 */
int mmm_alpha_objective(mmm_ctx* ctx, double alpha, double sum_Elnphi, int K, int V, double* val, double* grad);
/* Σ_d Σ_w n_w log(Σ_k props[k + K d] · phi[k V + v_w]) / Σ_d N_d over the documents with N_d > 0, CSR as in mmm_lda_create:
 * calculate_loglikelihood(X, θ, β) LDA.jl:174-188 (props = θ K x D, phi = β V x K column-major -- the same [k V + v]),
 * calculate_modality_loglikelihood(X, props, ϕ) MMCTM.jl:402-418, calculate_docmodality_loglikelihood (D = 1) MMCTM.jl:384-400. */
int mmm_mixture_loglik(mmm_ctx* ctx, int D, int K, int V, const int64_t* doc_ptr, const int32_t* term, const int32_t* count,
                       const double* props, const double* phi, double* ll);
/* The same with a topic-term probability that factorises over the I features of a term: Π_i phi[k ΣJ + Σ_{q<i} J_q + features[i V + v]]
 * (features 0-based, [i V + v]; phi in the IMMCTM gamma layout of one modality).  softmax = 1: calculate_modality_loglikelihood(X, η, ϕ,
 * features) -- IMMCTM.jl:362-407, props = softmax(eta[:, d]) (eta K x D); softmax = 0: eta holds the proportions themselves --
 * calculate_loglikelihood(X, features, θ, β) of ILDA.jl:203-231 (phi[k][i][j] = β[i][j, k]). */
int mmm_mixture_loglik_features(mmm_ctx* ctx, int D, int K, int V, int I, const int* J, const int32_t* features, const int64_t* doc_ptr,
                                const int32_t* term, const int32_t* count, const double* eta, int softmax, const double* phi, double* ll);

#ifdef __cplusplus
}
#endif
#endif /* MMMUSIG_H */
