// lda_estep.cuh -- the E-step kernels of lda.hip (included there, inside its anonymous namespace): k_lda_estep (CSR / padded rows, LDS slabs)
// and k_lda_estep_dense (rows of counts, statistics in registers).  LDA.jl:69-108.
__device__ __forceinline__ void lds_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// one chunk of L terms of a document group: phi_kw n_w into the accumulators and the wave's slab, and (LL) the
// log-likelihood numerator of the previous iteration.  __restrict__ tells the compiler that the slab atomics do not
// alias the table reads, so the reads of the following chunk can be issued ahead of them.
template <int KP, bool LL>
__device__ __forceinline__ void lda_chunk(const int2 tcv, const bool act, const int V, const double (&av)[KP], double (&acc)[KP],
                                          const double* __restrict__ sB, const double* __restrict__ sBeta,
                                          const double* __restrict__ myT, double* __restrict__ slab, double& ll_acc)
{
    const double n = (double)tcv.y;
    if (LL) {
        const double* bc = sBeta + tcv.x;
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int k = 0; k + 1 < KP; k += 2) { p0 = fma(myT[k], bc[k * V], p0); p1 = fma(myT[k + 1], bc[(k + 1) * V], p1); }
        if (KP & 1) p0 = fma(myT[KP - 1], bc[(KP - 1) * V], p0);
        ll_acc = fma(n, dev_log_pos(p0 + p1), ll_acc);                    // inactive lanes: n = 0
        __builtin_amdgcn_sched_barrier(0);                                  // keep the two halves' live ranges apart
    }
    const double* bcol = sB + tcv.x;
    if (KP >= 20) {
        // many topics: the products are formed twice (a second LDS read of the column) instead of being kept -- 64 registers
        // less, which is the difference between this build fitting its 256 and spilling
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k + 1 < KP; k += 2) { s0 = fma(av[k], bcol[k * V], s0); s1 = fma(av[k + 1], bcol[(k + 1) * V], s1); }
        const double r = act ? n * dev_rcp(s0 + s1) : 0.0;
        __builtin_amdgcn_sched_barrier(0);
        double* scol = slab + tcv.x;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const double x = av[k] * bcol[k * V] * r;
            acc[k] += x;
            if (act) unsafeAtomicAdd(&scol[k * V], x);
        }
        return;
    }
    double b[KP], s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < KP; ++k) b[k] = av[k] * bcol[k * V];
#pragma unroll
    for (int k = 0; k + 1 < KP; k += 2) { s0 += b[k]; s1 += b[k + 1]; }
    if (KP & 1) s0 += b[KP - 1];
    const double r = act ? n * dev_rcp(s0 + s1) : 0.0;
#pragma unroll
    for (int k = 0; k < KP; ++k) { b[k] *= r; acc[k] += b[k]; }        // phi_kw * n_w (padded topics: exact zeros)
    if (act) {
        double* scol = slab + tcv.x;
#pragma unroll
        for (int k = 0; k < KP; ++k) unsafeAtomicAdd(&scol[k * V], b[k]);
    }
}

// SINGLE: the grid covers every document with one step per wave (no step loop: 46 VGPRs less -> 3 waves per SIMD)
template <int KP, int L, bool LL, int VT, bool SINGLE>
__global__ __launch_bounds__(SINGLE ? kMaxWavesE * MMM_WAVE : 512, SINGLE ? 3 : 2) void k_lda_estep(EstepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int G = MMM_WAVE / L;                   // documents per wave step
    constexpr int PRE = (96 + L - 1) / L;             // chunks prefetched into registers (covers a 96-term document)
    MMM_STAMP(0);
    const int t = a.t;
    const int stop = a.ctl->stop;                     // consumed after the first prologue (its latency is hidden)
    const double* __restrict__ gam = a.gamma.s[t % 3];
    const double* __restrict__ gprev = a.gamma.s[(t + 2) % 3];
    double* __restrict__ gnext = a.gamma.s[(t + 1) % 3];
    double* __restrict__ Eln = a.Elntheta.s[t % 3];
    const double* __restrict__ eB = a.expElnbeta.s[(t + 2) % 3];
    const double* __restrict__ bprev = a.beta.s[(t + 2) % 3];

    const int K = a.c.K, D = a.c.D;
    const int V = VT ? VT : a.c.V;                    // VT != 0: row stride known at compile time (immediate LDS offsets)
    const int NW = blockDim.x >> 6;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane / L, l = lane % L;
    double* sB = smem;                                   // [KP][V] exp(Elnbeta_{t-1})
    double* sBeta = sB + (size_t)KP * V;                 // [KP][V] beta_{t-1}
    double* sSlab = sBeta + (size_t)KP * V;              // [NW][KP][V]
    double* sA = sSlab + (size_t)NW * KP * V;            // [NW][G][KP]
    double* sT = sA + (size_t)NW * G * KP;               // [NW][G][KP]
    double* slab = sSlab + (size_t)wid * KP * V;
    double* myA = sA + ((size_t)wid * G + g) * KP;
    double* myT = sT + ((size_t)wid * G + g) * KP;
    const int stride = gridDim.x * NW * G;
    int base = (blockIdx.x * NW + wid) * G;
    double ll_acc = 0.0;

    // ---- document loads of the first step are issued before the tables are staged (latency overlap) -------------
    int d = base + g;
    bool valid = d < D;
    // ext: this pass's prologue (digamma, exp: 2 us of this kernel's 10 at BASELINE config 2, all of it on every wave's dependent chain)
    // has run beside the previous pass's reduction, off the critical path; the same functions on the same lanes, hence the same bits
    const bool ext = SINGLE && !LL && a.aexp != nullptr;
    double gk = ext ? ((valid && l < K) ? a.aexp[(size_t)d * K + l] : 0.0) : ((valid && l < K) ? gam[(size_t)d * K + l] : (l < K ? 1.0 : 0.0));
    double gp = (LL && valid && l < K) ? gprev[(size_t)d * K + l] : (l < K ? 1.0 : 0.0);
    // Single-step build over padded rows (c.ell: [D][V] (term,count), (-1,0) past the document's end): the document's pairs are
    // addressed by d alone, so their loads leave with the gamma row instead of a memory round trip later (doc_ptr -> tc), every
    // document has the same V / L chunks (static register indices, no per-step shuffles)
    const bool drows = V <= PRE * L && (a.c.dense != nullptr || a.c.dense16 != nullptr);      // rows of counts: term = slot, 4 or 2 bytes per slot
    const bool rows = drows || (V <= PRE * L && a.c.ell != nullptr);      // (the grid-stride build requests the next step's row a step ahead)
    int64_t start = (valid && !rows) ? a.c.doc_ptr[d] : 0;
    int W = (valid && !rows) ? (int)(a.c.doc_ptr[d + 1] - start) : 0;
    // SINGLE: the table stays in registers (<= 5 entries per thread: KP*V <= 12 * 96, >= 4 waves) until just before the barrier, so
    // that the prologue arithmetic below runs while these loads are in flight instead of after them
    constexpr int TB = KP <= 10 ? 4 : 5;
    double tb[TB];
    if (SINGLE) {
#pragma unroll
        for (int q = 0; q < TB; ++q) { const int i = tid + q * (int)blockDim.x; tb[q] = (i < K * V) ? eB[i] : 0.0; }
    }
    for (int i = tid; i < NW * KP * V; i += blockDim.x) sSlab[i] = 0.0;
    for (int i = tid; i < KP * V; i += blockDim.x) {
        if (!SINGLE) sB[i] = (i < K * V) ? eB[i] : 0.0;
        if (LL) sBeta[i] = (i < K * V) ? bprev[i] : 0.0;
    }
    MMM_STAMP(1);

    // Grid-stride build: a two-deep software pipeline over the wave's steps.  The CSR offsets of step i+2 and the (term,count)
    // pairs + gamma row of step i+1 are requested while step i computes, so that no step starts with the two dependent memory
    // round trips doc_ptr -> tc (at 160k-640k documents they were ~70 % of a step: the SIMDs ran at 30 % VALU utilisation).
    int d1 = 0; bool valid1 = false; int64_t start1 = 0; int W1 = 0;
    if (!SINGLE) {
        d1 = base + stride + g; valid1 = (base + stride < D) && d1 < D;
        start1 = (valid1 && !rows) ? a.c.doc_ptr[d1] : 0;
        W1 = (valid1 && !rows) ? (int)(a.c.doc_ptr[d1 + 1] - start1) : 0;
    }
    int2 tcp[PRE];                           // (term,count) of the first PRE chunks of the current step
    bool first = true;
    for (;;) {
        // ---- groups start at rotated chunks so that the G documents of a wave instruction touch different term ranges of the slab
        const int NCHR = VT ? (VT + L - 1) / L : (V + L - 1) / L;      // chunks of a padded row (<= PRE)
        const int nch = rows ? NCHR : (W + L - 1) / L;
        const int rot = nch > 0 ? g % nch : 0;
        const int2* __restrict__ tcd = a.c.tc + start;
        int nchmax = nch;
        if (!rows) {
            if (G >= 2) nchmax = max(nchmax, __shfl_xor(nchmax, 32, MMM_WAVE));
            if (G >= 4) nchmax = max(nchmax, __shfl_xor(nchmax, 16, MMM_WAVE));
            nchmax = __builtin_amdgcn_readfirstlane(nchmax);
        }
        if (drows && (SINGLE || first)) {
            const int* __restrict__ row = a.c.dense + (size_t)(valid ? d : 0) * a.c.Vp;
            const unsigned short* __restrict__ row16 = a.c.dense16 + (size_t)(valid ? d : 0) * a.c.Vp;
            const bool h16 = a.c.dense16 != nullptr;
            const int slp = a.c.Vp >> 4;
            if (L == 16 && h16) {      // one 16-byte load instead of one 2-byte load per chunk (six loads whose last waited for the first five)
                const unsigned* __restrict__ r32 = (const unsigned*)(row16 + (size_t)l * slp);
                const unsigned w0 = r32[0], w1 = r32[1], w2 = r32[2], w3 = r32[3];
#pragma unroll
                for (int j = 0; j < PRE; ++j) {
                    int c = j + rot; if (c >= NCHR) c -= NCHR;
                    const int w = c * L + l;
                    const bool in = valid && j < NCHR && w < V;
                    const int n = in ? row16_count(w0, w1, w2, w3, c) : 0;
                    tcp[j] = make_int2(n > 0 ? w : -1, n);
                }
            } else
#pragma unroll
            for (int j = 0; j < PRE; ++j) {
                int c = j + rot; if (c >= NCHR) c -= NCHR;
                const int w = c * L + l;
                const bool in = valid && j < NCHR && w < V;
                const int n = in ? (h16 ? (int)row16[row_slot(w, slp)] : row[row_slot(w, slp)]) : 0;
                tcp[j] = make_int2(n > 0 ? w : -1, n);
            }
        } else if (rows && !drows && (SINGLE || first)) {
            const int2* __restrict__ row = a.c.ell + (size_t)(valid ? d : 0) * V;
#pragma unroll
            for (int j = 0; j < PRE; ++j) {
                int c = j + rot; if (c >= NCHR) c -= NCHR;
                const int w = c * L + l;
                tcp[j] = (valid && j < NCHR && w < V) ? row[w] : make_int2(-1, 0);
            }
        } else if (!rows && (SINGLE || first)) {               // first step: loads issued before the prologue math (later steps: requested a step ahead)
#pragma unroll
            for (int j = 0; j < PRE; ++j) {
                int c = j + rot; if (c >= nch) c -= nch;
                const int w = c * L + l;
                tcp[j] = ((j < nch) && (w < W)) ? tcd[w] : make_int2(-1, 0);
            }
        }
        // ---- Elntheta (LDA.jl:78-80), a_k = exp(Elntheta_k), theta_{t-1} (LDA.jl:92-94) ------------------------------
        double el = 0.0;
        if (ext) { if (l < KP) myA[l] = gk; }
        else {
            const double S = group_sum<L>(gk);
            const double ps = dev_digamma_pos(l < K ? gk : S);        // lane K of the group holds psi(S)
            const double psS = __shfl(ps, g * L + K, MMM_WAVE);
            el = ps - psS;
            if (l < KP) myA[l] = (l < K) ? ar_exp(el) : 0.0;
        }
        if (LL) {
            const double Sp = group_sum<L>(gp);
            if (l < KP) myT[l] = (l < K) ? gp / Sp : 0.0;
        }
        if (first) {
            if (stop) return;            // a previous pass met the stopping rule: this launch must not touch the state
            if (SINGLE) {
#pragma unroll
                for (int q = 0; q < TB; ++q) { const int i = tid + q * (int)blockDim.x; if (i < KP * V) sB[i] = tb[q]; }
            }
            __syncthreads();
            first = false;
            MMM_STAMP(2);
        } else lds_wave_sync();
        if (!ext && valid && l < K) Eln[(size_t)d * K + l] = el;
        MMM_STAMP(3);
        // ---- requests of the next two steps (grid-stride build) ---------------------------------------------------------
        int2 tcn[PRE];
        double gkn = 0.0, gpn = 0.0;
        int d2 = 0; bool valid2 = false; int64_t start2 = 0; int W2 = 0;
        const bool more = !SINGLE && base + stride < D;
        if (more) {
            const int nch1 = rows ? NCHR : (W1 + L - 1) / L;
            const int rot1 = nch1 > 0 ? g % nch1 : 0;
            const int2* __restrict__ tcd1 = a.c.tc + start1;
            if (drows) {
                const int* __restrict__ row = a.c.dense + (size_t)(valid1 ? d1 : 0) * a.c.Vp;
                const unsigned short* __restrict__ row16 = a.c.dense16 + (size_t)(valid1 ? d1 : 0) * a.c.Vp;
                const bool h16 = a.c.dense16 != nullptr;
                const int slp = a.c.Vp >> 4;
                if (L == 16 && h16) {
                    const unsigned* __restrict__ r32 = (const unsigned*)(row16 + (size_t)l * slp);
                    const unsigned w0 = r32[0], w1 = r32[1], w2 = r32[2], w3 = r32[3];
#pragma unroll
                    for (int j = 0; j < PRE; ++j) {
                        int c = j + rot1; if (c >= NCHR) c -= NCHR;
                        const int w = c * L + l;
                        const bool in = valid1 && j < NCHR && w < V;
                        const int n = in ? row16_count(w0, w1, w2, w3, c) : 0;
                        tcn[j] = make_int2(n > 0 ? w : -1, n);
                    }
                } else
#pragma unroll
                for (int j = 0; j < PRE; ++j) {
                    int c = j + rot1; if (c >= NCHR) c -= NCHR;
                    const int w = c * L + l;
                    const bool in = valid1 && j < NCHR && w < V;
                    const int n = in ? (h16 ? (int)row16[row_slot(w, slp)] : row[row_slot(w, slp)]) : 0;
                    tcn[j] = make_int2(n > 0 ? w : -1, n);
                }
            } else if (rows) {
                const int2* __restrict__ row = a.c.ell + (size_t)(valid1 ? d1 : 0) * V;
#pragma unroll
                for (int j = 0; j < PRE; ++j) {
                    int c = j + rot1; if (c >= NCHR) c -= NCHR;
                    const int w = c * L + l;
                    tcn[j] = (valid1 && j < NCHR && w < V) ? row[w] : make_int2(-1, 0);
                }
            } else {
#pragma unroll
            for (int j = 0; j < PRE; ++j) {
                int c = j + rot1; if (c >= nch1) c -= nch1;
                const int w = c * L + l;
                tcn[j] = ((j < nch1) && (w < W1)) ? tcd1[w] : make_int2(-1, 0);
            }
            }
            gkn = (valid1 && l < K) ? gam[(size_t)d1 * K + l] : (l < K ? 1.0 : 0.0);
            gpn = (LL && valid1 && l < K) ? gprev[(size_t)d1 * K + l] : (l < K ? 1.0 : 0.0);
            d2 = base + 2 * stride + g; valid2 = (base + 2 * stride < D) && d2 < D;
            start2 = (valid2 && !rows) ? a.c.doc_ptr[d2] : 0;
            W2 = (valid2 && !rows) ? (int)(a.c.doc_ptr[d2 + 1] - start2) : 0;
        }
        {
            double av[KP], acc[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) { av[k] = myA[k]; acc[k] = 0.0; }
            if (rows) {
#pragma unroll
                for (int j = 0; j < PRE; ++j) {
                    if (j < NCHR) {
                        int2 tcv = tcp[j];
                        const bool act = tcv.x >= 0;
                        tcv.x = act ? tcv.x : 0;
                        lda_chunk<KP, LL>(tcv, act, V, av, acc, sB, sBeta, myT, slab, ll_acc);
                    }
                }
            } else
#pragma unroll 2
            for (int j = 0; j < nchmax; ++j) {
                int2 tcv = tcp[0];
#pragma unroll
                for (int q = 1; q < PRE; ++q) tcv = (j == q) ? tcp[q] : tcv;     // register select (static indices only)
                if (j >= PRE) {
                    int c = j + rot; if (c >= nch) c -= nch;
                    const int w = c * L + l;
                    tcv = ((j < nch) && (w < W)) ? tcd[w] : make_int2(-1, 0);
                }
                const bool act = tcv.x >= 0;
                tcv.x = act ? tcv.x : 0;
                lda_chunk<KP, LL>(tcv, act, V, av, acc, sB, sBeta, myT, slab, ll_acc);
            }
            MMM_STAMP(4);
            // ---- gamma_{t+1} = alpha + sum_w phi_kw n_w (LDA.jl:83-87 of the next pass) ---------------------------
            double mine = 0.0;
#pragma unroll
            for (int k = 0; k < KP; ++k) { const double tot = group_sum<L>(acc[k]); if (l == k) mine = tot; }
            if (valid && l < K) gnext[(size_t)d * K + l] = a.c.alpha + mine;
        }
        MMM_STAMP(5);
        if (SINGLE) break;
        base += stride;
        if (base >= D) break;
        // ---- the next step's operands were requested above ------------------------------------------------------------
        d = d1; valid = valid1; gk = gkn; gp = gpn; start = start1; W = W1;
#pragma unroll
        for (int j = 0; j < PRE; ++j) tcp[j] = tcn[j];
        d1 = d2; valid1 = valid2; start1 = start2; W1 = W2;
        lds_wave_sync();
    }
    MMM_STAMP(6);
    // ---- block epilogue: slabs -> one partial; ll partial ------------------------------------------------------
    if (LL) ll_acc = wave_sum(ll_acc);
    __syncthreads();
    if (LL && lane == 0) sA[wid] = ll_acc;      // sA is free now
    double* out = a.partial + (size_t)blockIdx.x * K * a.pstride;
    for (int i = tid; i < K * V; i += blockDim.x) {
        double v8[kMaxWavesE];
#pragma unroll
        for (int w = 0; w < kMaxWavesE; ++w) v8[w] = (w < NW) ? sSlab[(size_t)w * KP * V + i] : 0.0;
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kMaxWavesE; ++w) s += v8[w];
        out[a.pstride == V ? i : (i / V) * a.pstride + i % V] = s;
    }
    if (LL) {
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int w = 0; w < NW; ++w) s += sA[w];
            a.llpart[blockIdx.x] = s;
        }
    }
    MMM_STAMP(7);
}


// ---- dense-row E-step (large corpora over a small vocabulary) ---------------------------------------------------------------------
// The corpora this model is used on are dense: mutation catalogues over the 96 SNV channels list nearly every channel in every sample
// (data/brca-eu_snv_counts.tsv: 53,559 of 53,760 entries).  For such a corpus the CSR sweep of k_lda_estep spends its time in LDS: per
// nonzero an 80-byte table column read and an 80-byte read-modify-write of the wave's slab (PMC: LDS pipe, not VALU, bounds the chunk
// loop).  Here a document is a row of Vp = 16 SL counts (zeros where a term is absent); lane l of a 16-lane document group owns the terms
// l, 16 + l, ..., the same ones in every document it meets, so the statistics sum_d phi_kv n_dv of its terms stay in REGISTERS for the
// whole launch (SL * KP doubles per lane) and reach the slab once, at the end.  Per term slot: a conflict-free 16-byte-per-lane read
// of the term-major table and 2 KP + 16 f64 instructions; no atomics in the loop.  HBM per document: 4 Vp bytes of counts instead of
// 8 bytes per nonzero.  Same formulas as k_lda_estep (LDA.jl:69-108); the sums are associated per lane, then lanes, waves, blocks.
template <class T> __device__ __forceinline__ T* at_byte(T* base, unsigned off) { return (T*)((char*)base + off); }   // uniform base + 32-bit lane offset: one VGPR per address
typedef unsigned short mmm_us2 __attribute__((ext_vector_type(2)));

template <int KP, int SL, bool C16>
__global__ __launch_bounds__(512, 2) void k_lda_estep_dense(EstepArgs a, const int* __restrict__ cnt, const unsigned short* __restrict__ cnt16)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int L = 16, G = MMM_WAVE / L, Vp = L * SL;
    MMM_STAMP(0);
    const int t = a.t;
    const int stop = a.ctl->stop;
    const double* __restrict__ gam = a.gamma.s[t % 3];
    double* __restrict__ gnext = a.gamma.s[(t + 1) % 3];
    double* __restrict__ Eln = a.Elntheta.s[t % 3];
    const double* __restrict__ eB = a.expElnbeta.s[(t + 2) % 3];
    const int K = a.c.K, D = a.c.D, V = a.c.V;
    const int NW = blockDim.x >> 6;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane / L, l = lane % L;
    double* sT = smem;                                   // [Vp][KP] exp(Elnbeta_{t-1}), term-major; rows v >= V hold 1 (their counts are 0)
    double* sSlab = sT + (size_t)Vp * KP;                // [NW][Vp][KP], term-major like the table (written once, in the epilogue)
    double* sA = sSlab + (size_t)NW * Vp * KP;           // [NW][G][KP]
    double* sR = sA + (size_t)NW * G * KP;               // [NW][64][KP] gamma sums, lane-major
    double* slab = sSlab + (size_t)wid * Vp * KP;
    double* myA = sA + ((size_t)wid * G + g) * KP;
    double* myR = sR + (size_t)wid * MMM_WAVE * KP;
    const int stride = gridDim.x * NW * G;
    int base = (blockIdx.x * NW + wid) * G;
    // exp / log tables of the prologue (mmm_arith.h: ar_exp_tab, ar_digamma_pos_tab -- a third of this kernel's vector work is the K + 1
    // digammas and K exps per document; no division in exp, 20 instructions fewer in the log)
    __shared__ __attribute__((aligned(16))) double sTabs[MMM_EXPTAB_N + MMM_LOGTAB_N];
    for (int i = tid; i < MMM_EXPTAB_N + MMM_LOGTAB_N; i += blockDim.x) sTabs[i] = i < MMM_EXPTAB_N ? g_mmm_exptab[i] : g_mmm_logtab[i - MMM_EXPTAB_N];

    // The next step's gamma row and counts are requested a step ahead and must stay in flight across the step: nothing between a request and
    // its use may wait for memory (vmcnt counts in order, so ONE scratch reload in the loop waits for every load before it -- the build
    // that spilled 12 registers exposed the HBM round trip in every step: 640k documents 280 us at 53 % of its own issue time).  So: loads
    // are unconditional (a clamped document index; masks are applied when the values are used), 16-bit counts land in register halves
    // (SL / 2 registers), addresses are a uniform base + one 32-bit offset per lane (D K 8 and D Vp 4 bytes < 4 GB, checked at create).
    constexpr int NC = C16 ? (SL + 1) / 2 : SL;
    int d = base + g;
    bool valid = d < D;
    const int lk = l < K ? l : K - 1;
    unsigned dl = valid ? (unsigned)d : 0u;
    double gk = *at_byte(gam, (dl * (unsigned)K + lk) * 8u);
    gk = (valid && l < K) ? gk : (l < K ? 1.0 : 0.0);
    constexpr int SLs = C16 ? 2 * NC : SL;                  // slots a lane owns in a stored row
    unsigned c[NC], cn[NC];
    auto request = [&](unsigned* o, unsigned dd) {          // the lane's part of the row: NC consecutive 32-bit words, one load
        const unsigned* row = C16 ? at_byte((const unsigned*)cnt16, (dd * (unsigned)(16 * SLs) + l * SLs) * 2u)
                                  : at_byte((const unsigned*)cnt, (dd * (unsigned)(16 * SLs) + l * SLs) * 4u);
#pragma unroll
        for (int j = 0; j < NC; ++j) o[j] = row[j];
    };
    auto take = [&](const unsigned* raw, bool ok) {
#pragma unroll
        for (int j = 0; j < NC; ++j) c[j] = ok ? raw[j] : 0u;
    };
    request(cn, dl);
    take(cn, valid);
    for (int i = tid; i < Vp * KP; i += blockDim.x) {
        const int v = i / KP, k = i % KP;
        sT[i] = (k < K) ? (v < V ? eB[(size_t)k * V + v] : 1.0) : 0.0;
    }
    __syncthreads();          // the function tables are read by the first step's prologue
    double st[SL][KP];
#pragma unroll
    for (int q = 0; q < SL; ++q)
#pragma unroll
        for (int k = 0; k < KP; ++k) st[q][k] = 0.0;
    bool first = true;
    for (;;) {
        // ---- the next step's gamma row and counts are requested before this step's term phase
        const int dn = d + stride;
        const bool more = base + stride < D, validn = more && dn < D;
        const unsigned dnl = validn ? (unsigned)dn : dl;
        // ---- Elntheta (LDA.jl:78-80), a_k = exp(Elntheta_k)
        const double S = group_sum<L>(gk);
        const double ps = ar_digamma_pos_tab(l < K ? gk : S, sTabs + MMM_EXPTAB_N);        // lane K of the group holds psi(S)
        const double psS = __shfl(ps, g * L + K, MMM_WAVE);
        const double el = ps - psS;
        const double ak = (l < K) ? ar_exp_tab(el, sTabs) : 0.0;
        if (l < KP) myA[l] = ak;
        // (requested here, after the prologue: its polynomial constants overflow the scalar registers and one is reloaded from scratch in there)
        const double gkn = *at_byte(gam, (dnl * (unsigned)K + lk) * 8u);
        request(cn, dnl);
        if (first) {
            if (stop) return;            // a previous pass met the stopping rule: this launch must not touch the state
            __syncthreads();
            first = false;
            MMM_STAMP(1);
        } else lds_wave_sync();
        if (valid && l < K) *at_byte(Eln, (dl * (unsigned)K + l) * 8u) = el;
        double acc[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) acc[k] = 0.0;
        // ---- phi_kv n_v (LDA.jl:92-106) for the lane's SL terms.  (Tried at K = 10, V = 96, 640k documents: a_k re-read from LDS in every slot,
        // no spilled register instead of 12: 330 vs 307 us; the next slot's table row requested a slot ahead, 62 spilled: 509 us.)
        double av[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) av[k] = myA[k];
#pragma unroll
        for (int q = 0; q < SL; ++q) {
            const double* tb = sT + (size_t)(q * L + l) * KP;
            // three fused multiply-adds per (term slot, topic): the normaliser s = sum_k a_k B_kv, the lane's gamma sums WITHOUT their factor
            // a_k (it is the document's, applied once after the lanes' sums have met) and the statistics WITHOUT their factor B_kv (it is
            // the term's, the same for every document, applied once when the registers reach the slab)
            double b[KP], s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int k = 0; k < KP; ++k) b[k] = tb[k];
#pragma unroll
            for (int k = 0; k + 1 < KP; k += 2) { s0 = fma(av[k], b[k], s0); s1 = fma(av[k + 1], b[k + 1], s1); }
            if (KP & 1) s0 = fma(av[KP - 1], b[KP - 1], s0);
            // a slot without mass must not see 0 x rcp(0) = NaN: with tiny priors the normaliser of a never-observed term underflows to 0.
            // (v_max with the smallest normal: the bits of every other quotient are unchanged; a select on the count costs 150 spilled registers here)
            const unsigned cq = C16 ? ((q & 1) ? c[q / 2] >> 16 : c[q / 2] & 0xffffu) : c[q];
            const double r = (double)cq * dev_rcp(dev_max_raw(s0 + s1, 2.2250738585072014e-308));
#pragma unroll
            for (int k = 0; k < KP; ++k) { acc[k] = fma(b[k], r, acc[k]); st[q][k] = fma(av[k], r, st[q][k]); }
            // one slot at a time, its statistics updated here (left alone the compiler sinks the SL KP updates to the end of the step and keeps
            // every slot's products alive until then: 190 spilled registers)
#pragma unroll
            for (int k = 0; k < KP; ++k) asm volatile("" : "+v"(st[q][k]));
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        MMM_STAMP(2);
        // ---- the requested values are taken over HERE, before the step's last store: the compiler prices a wait for loads as if the
        // (lane-conditional) stores after them had not been issued, i.e. as vmcnt(0) -- placed after the gamma store below it waited for
        // that store's round trip in every step
        const bool valid_now = valid;
        const unsigned dl_now = dl;
        // (the empty statements are the first use of the loaded registers and cannot move above the term phase's own)
        double gk_next = gkn;
        asm volatile("" : "+v"(gk_next) :: "memory");
#pragma unroll
        for (int j = 0; j < NC; ++j) asm volatile("" : "+v"(cn[j]) :: "memory");
        gk_next = (validn && l < K) ? gk_next : (l < K ? 1.0 : 0.0);
        take(cn, validn);
#pragma unroll
        for (int j = 0; j < NC; ++j) asm volatile("" : "+v"(c[j]));
        __builtin_amdgcn_sched_barrier(0);
        // ---- gamma_{t+1} = alpha + sum_v phi_kv n_v: the lanes' sums meet in LDS, lane k of the group adds its column
#pragma unroll
        for (int k = 0; k < KP; ++k) myR[(size_t)lane * KP + k] = acc[k];
        lds_wave_sync();
        if (l < K) {
            const double* col = myR + (size_t)(g * L) * KP + l;
            double r0 = col[0], r1 = col[KP], r2 = col[2 * KP], r3 = col[3 * KP];
#pragma unroll
            for (int j = 4; j < L; j += 4) { r0 += col[j * KP]; r1 += col[(j + 1) * KP]; r2 += col[(j + 2) * KP]; r3 += col[(j + 3) * KP]; }
            if (valid_now) *at_byte(gnext, (dl_now * (unsigned)K + l) * 8u) = fma(ak, (r0 + r1) + (r2 + r3), a.c.alpha);
        }
        base += stride;
        if (base >= D) break;
        d = dn; valid = validn; dl = dnl; gk = gk_next;
        lds_wave_sync();
    }
    MMM_STAMP(3);
    // ---- the wave's statistics: the four document groups' registers are added across the rows of the wave (rows_sum4: (g0 + g2) + (g1 + g3),
    // no LDS), multiplied by the term's table entry once, and the first group's lanes store them -- the slab is term-major with padded
    // bounds like the table, written once (no zero fill, no read-modify-write; 16-byte pairs at compile-time offsets).  (One group at a
    // time through LDS with run-time bounds, every entry its own round trip: 19 of the 27 us of a 15k-document launch,
    // tools/diag_dense_stamps.py; batched per term slot: 6.4.)
#pragma unroll
    for (int q = 0; q < SL; ++q) {
        double* sl = slab + (size_t)(q * L + l) * KP;
        const double* tb = sT + (size_t)(q * L + l) * KP;
        double t[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) t[k] = rows_sum4(st[q][k]) * tb[k];
        if (g == 0) {
#pragma unroll
            for (int k = 0; k < KP; ++k) sl[k] = t[k];
        }
    }
    MMM_STAMP(4);
    __syncthreads();
    MMM_STAMP(5);
    double* out = a.partial + (size_t)blockIdx.x * K * a.pstride;
    for (int i = tid; i < K * V; i += blockDim.x) {
        const int kk = i / V, v = i - kk * V;
        double s = 0.0;
        for (int w = 0; w < NW; ++w) s += sSlab[((size_t)w * Vp + v) * KP + kk];
        out[a.pstride == V ? i : kk * a.pstride + v] = s;
    }
    MMM_STAMP(6);
    MMM_STAMP(7);
}
