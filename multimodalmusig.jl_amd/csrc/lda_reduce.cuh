// lda_reduce.cuh -- what follows the E-step kernel in a pass of lda.hip (included there, inside its anonymous namespace): slab reduction, the
// lagged log-likelihood blocks, the merged reduce + ll + M-step launch, the split M-step kernels, the pass tail.  LDA.jl:96-112, 174-224.
// ---- slab reduction + log-likelihood / stopping rule ------------------------------------------------------------
struct ReduceArgs {
    const double* partial; const double* llpart; int nslab; int VK;
    double* stats;         // out: [VK] summed lambda statistics of pass t, [VK] ll numerator of pass t-1
    LdaCtl* ctl;
    int t;                 // this pass (host count)
    double Nglobal, tol;
    double* ll_hist;
    int do_ll, conv_base, run_tail;
    // several GPUs with the mailboxes up: k_lda_reduce sends its entries to the peers as it produces them and k_lda_mstep sums
    // the contributions in rank order as it consumes them -- the all-reduce costs no launch of its own
    int p2p; unsigned int p2p_seq; P2PArgs px;
    // the log-likelihood of pass t-1 evaluated by extra blocks of the reduce launch (k_lda_reduce_ll) instead of inside the
    // E-step kernel: per-block numerators in llpart2[n_ll], summed (and exchanged) by the pass-tail block
    const double* llpart2; int n_ll, ll_in_k2;
    // RCCL transport: the numerator has to sit in stats[VK] before ncclAllReduce, so wave 1 of reduce block 0 collects the ll blocks'
    // numerators inside the reduce launch, through seq-tagged cells (as k_lda_reduce_ll_mstep does); the tail then only finishes
    unsigned long long* ll_cells; unsigned int ll_seq;
};

// ll_{t-1}, the convergence test of common.jl:53-56 after > 10 values (LDA.jl:215) and t += 1 (one thread)
__device__ void lda_pass_tail(const ReduceArgs& r)
{
    int stop = 0;
    if (r.do_ll) {
        const int n = r.ctl->n_hist;
        const double ll = r.stats[r.VK] / r.Nglobal;
        r.ll_hist[n] = ll;
        r.ctl->n_hist = n + 1;
        if (n + 1 - r.conv_base > 10) {
            const double prev = r.ll_hist[n - 1];
            if (fabs(prev - ll) / fabs(ll) < r.tol) { stop = 1; r.ctl->stop = 1; r.ctl->stop_iter = r.t - 1; }
        }
    }
    if (!stop) r.ctl->t = r.t;     // on convergence at t-1 the state of pass t is discarded
    r.ctl->ticket = 0;
}

// the pass-tail block (one wave) of the M-step launches: finishes the ll numerator of pass t-1 -- sum of the k_lda_reduce_ll
// partials and/or the peers' share -- and runs lda_pass_tail
template <bool P2P>
__device__ __forceinline__ void lda_tail_block(const ReduceArgs& r, int lane)
{
    double v = 0.0;
    const bool from_parts = r.ll_in_k2 && !r.ll_cells;
    if (from_parts && r.do_ll) {
        for (int i = lane; i < r.n_ll; i += 64) v += r.llpart2[i];
        v = wave_sum(v);
    }
    if (lane != 0) return;
    if (from_parts) {
        if (r.do_ll) {
            if (P2P && r.p2p) { p2p_send(r.px, r.p2p_seq, r.VK, v); v = p2p_recv_sum(r.px, r.p2p_seq, r.VK, v); }
            r.stats[r.VK] = v;
        }
    } else if (P2P && r.p2p) r.stats[r.VK] = p2p_recv_sum(r.px, r.p2p_seq, r.VK, r.stats[r.VK]);
    lda_pass_tail(r);
}

// log-likelihood numerator of pass t-1 (LDA.jl:174-188 with theta_{t-1} = gamma_{t-1} / sum, beta_{t-1}) for the documents of
// "ll block" lb of nlb, by a block of 16 waves laid out like k_lda_reduce's (16 x 64 threads): L lanes per document (as in the
// E-step), 64/L documents per wave step, beta staged in LDS -- the ll half of the E-step's chunk loop, moved out of it.
template <int KP, int L>
__device__ void lda_ll_block(const LdaDev& c, const double* __restrict__ gprev, const double* __restrict__ bprev, double* llpart2, int lb, int nlb,
                             double* smem, unsigned long long* cell = nullptr, unsigned int seq = 0)
{
    __shared__ double s_w[16];
    const int tid = threadIdx.y * 16 + threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int G = MMM_WAVE / L;
    const int g = lane / L, l = lane % L;
    const int K = c.K, V = c.V, D = c.D;
    double* sBeta = smem;
    double* myT = smem + (size_t)KP * V + ((size_t)wid * G + g) * KP;
    double* sLog = smem + (size_t)KP * V + (size_t)64 * KP;         // [256] the log table (dev_log_tab)
    // the first step's document loads go out before the table is staged (as in the E-step kernel).  (Splitting a document
    // group's chunks over 2 or 4 waves -- more, lighter blocks on the CUs the reduction leaves idle -- was slower: 29.7 / 33.6
    // vs 26.6 us per iteration; the launch is bound by block dispatch and table staging, not by the sweep's arithmetic.)
    // wave w of block lb is wave slot w * nlb + lb: the documents fill wave 0 of every block, then wave 1, ... -- a corpus of fewer
    // than 64 nlb documents leaves every block the same number of busy waves (the sweep is issue-bound per CU: 157 blocks of 16 busy
    // waves were 0.6 us slower at BASELINE config 2 than 192 blocks of 13)
    // (corpora of more than one step per wave keep a block's waves on neighbouring documents)
    const int wslot = ((int64_t)nlb * 16 * G >= (int64_t)D) ? wid * nlb + lb : lb * 16 + wid, nslots = nlb * 16;
    int base = wslot * G;
    int d = base + g;
    bool valid = d < D;
    double gp = (valid && l < K) ? gprev[(size_t)d * K + l] : (l < K ? 1.0 : 0.0);
    constexpr int PRE = 128 / L;          // padded rows: every chunk of the document is requested up front, no doc_ptr needed
    const bool dense = L == 16 && (c.dense != nullptr || c.dense16 != nullptr);       // rows of counts: term = slot index, 4 or 2 bytes per slot, table columns read in lane order
    const bool h16 = c.dense16 != nullptr;
    const bool ell = dense || c.ell != nullptr;
    int2 pre[PRE];
    // 16-bit lane-major rows: the lane's <= 8 slots are the first words of ONE 16-byte load (the rows are allocated with 16 bytes to spare),
    // and in the loop below the next step's gamma row and counts are requested while this step computes (the rules of k_lda_estep_dense)
    const bool fast = L == 16 && dense && h16 && (int64_t)D * K * 8 < ((int64_t)1 << 32) && (int64_t)D * c.Vp * 2 < ((int64_t)1 << 32);
    unsigned wq[4] = {0u, 0u, 0u, 0u};
    const int lk = l < K ? l : K - 1;
    unsigned dl = valid ? (unsigned)d : 0u;
    if (fast) {
        const unsigned* row = at_byte((const unsigned*)c.dense16, (dl * 16u + (unsigned)l) * (unsigned)(c.Vp >> 4) * 2u);
#pragma unroll
        for (int j = 0; j < 4; ++j) wq[j] = row[j];
    } else if (dense) {
        const int* __restrict__ row = c.dense + (size_t)(valid ? d : 0) * c.Vp;
        const unsigned short* __restrict__ row16 = c.dense16 + (size_t)(valid ? d : 0) * c.Vp;
#pragma unroll
        for (int j = 0; j < PRE; ++j) pre[j] = (valid && j * L + l < V) ? make_int2(j * L + l, h16 ? (int)row16[row_slot(j * L + l, c.Vp >> 4)] : row[row_slot(j * L + l, c.Vp >> 4)]) : make_int2(-1, 0);
    } else if (ell) {
        const int2* __restrict__ row = c.ell + (size_t)(valid ? d : 0) * V;
#pragma unroll
        for (int j = 0; j < PRE; ++j) pre[j] = (valid && j * L + l < V) ? row[j * L + l] : make_int2(-1, 0);
    }
    int64_t start = (!ell && valid) ? c.doc_ptr[d] : 0;
    int W = (!ell && valid) ? (int)(c.doc_ptr[d + 1] - start) : 0;
    // term-major copy [v][KP] of beta_{t-1}: a lane reads the KP entries of its term as 16-byte pairs at immediate offsets (lane stride
    // 8 KP bytes: the 16 lanes of a document group cover the banks once), instead of KP reads with an address computed for each
    for (int i = tid; i < KP * V; i += 1024) { const int v = i / KP, k = i - v * KP; sBeta[i] = (k < K) ? bprev[(size_t)k * V + v] : 0.0; }
    if (tid < MMM_LOGTAB_N) sLog[tid] = g_mmm_logtab[tid];
    MMM_RSTAMP(lb == 0 && tid == 0, 20);       // own loads (gamma row, document row, table entries) have arrived
    __syncthreads();
    MMM_RSTAMP(lb == 0 && tid == 0, 21);       // tables staged by all waves
    double acc = 0.0;
    if (fast) {
        const int nch = (V + L - 1) / L;
        const int stride = nslots * G;
        const unsigned slp2 = (unsigned)(c.Vp >> 4) * 2u;
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = valid ? wq[j] : 0u;
        for (; base < D; base += stride) {
            const int dn = d + stride;
            const bool validn = base + stride < D && dn < D;
            const unsigned dnl = validn ? (unsigned)dn : dl;
            const double Sp = group_sum<L>(gp);
            lds_wave_sync();
            if (l < KP) myT[l] = (l < K) ? dev_div(gp, Sp) : 0.0;      // (the bits of gp / Sp: both are normal numbers)
            lds_wave_sync();
            double tv[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) tv[k] = myT[k];
            // the next step's values (unconditional loads, clamped indices; taken over after the chunks)
            double gpn = *at_byte(gprev, (dnl * (unsigned)K + (unsigned)lk) * 8u);
            {
                const unsigned* row = at_byte((const unsigned*)c.dense16, (dnl * 16u + (unsigned)l) * slp2);
#pragma unroll
                for (int j = 0; j < 4; ++j) wq[j] = row[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < nch) {
                    const unsigned cq = (j & 1) ? w[j / 2] >> 16 : w[j / 2] & 0xffffu;
                    const bool act = valid && j * L + l < V;
                    const double* bc = sBeta + (size_t)(act ? j * L + l : 0) * KP;
                    double p0 = 0.0, p1 = 0.0;
#pragma unroll
                    for (int k = 0; k + 1 < KP; k += 2) { p0 = fma(tv[k], bc[k], p0); p1 = fma(tv[k + 1], bc[k + 1], p1); }
                    if (KP & 1) p0 = fma(tv[KP - 1], bc[KP - 1], p0);
                    const double p = act ? p0 + p1 : 1.0;
                    acc = fma((double)cq, dev_log_tab(p, sLog), acc);
                }
            }
            asm volatile("" : "+v"(acc) :: "memory");
            asm volatile("" : "+v"(gpn) :: "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(wq[j]) :: "memory");
            d = dn; valid = validn; dl = dnl;
            gp = (valid && l < K) ? gpn : (l < K ? 1.0 : 0.0);
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = valid ? wq[j] : 0u;
        }
    } else if (ell) {
        const int nch = (V + L - 1) / L;
        for (; base < D; base += nslots * G) {
            if (base != wslot * G) {
                d = base + g; valid = d < D;
                gp = (valid && l < K) ? gprev[(size_t)d * K + l] : (l < K ? 1.0 : 0.0);
                if (dense) {
                    const int* __restrict__ row = c.dense + (size_t)(valid ? d : 0) * c.Vp;
                    const unsigned short* __restrict__ row16 = c.dense16 + (size_t)(valid ? d : 0) * c.Vp;
#pragma unroll
                    for (int j = 0; j < PRE; ++j) pre[j] = (valid && j * L + l < V) ? make_int2(j * L + l, h16 ? (int)row16[row_slot(j * L + l, c.Vp >> 4)] : row[row_slot(j * L + l, c.Vp >> 4)]) : make_int2(-1, 0);
                } else {
                    const int2* __restrict__ row = c.ell + (size_t)(valid ? d : 0) * V;
#pragma unroll
                    for (int j = 0; j < PRE; ++j) pre[j] = (valid && j * L + l < V) ? row[j * L + l] : make_int2(-1, 0);
                }
            }
            const double Sp = group_sum<L>(gp);
            lds_wave_sync();
            if (l < KP) myT[l] = (l < K) ? dev_div(gp, Sp) : 0.0;      // (the bits of gp / Sp: both are normal numbers)
            lds_wave_sync();
            double tv[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) tv[k] = myT[k];
#pragma unroll
            for (int j = 0; j < PRE; ++j) {
                if (j >= nch) break;
                const int2 t = pre[j];
                const bool act = t.x >= 0;
                const double* bc = sBeta + (size_t)(act ? t.x : 0) * KP;
                double p0 = 0.0, p1 = 0.0;
#pragma unroll
                for (int k = 0; k + 1 < KP; k += 2) { p0 = fma(tv[k], bc[k], p0); p1 = fma(tv[k + 1], bc[k + 1], p1); }
                if (KP & 1) p0 = fma(tv[KP - 1], bc[KP - 1], p0);
                const double p = act ? p0 + p1 : 1.0;
                acc = fma((double)t.y, dev_log_tab(p, sLog), acc);
            }
        }
    } else
    for (; base < D; base += nslots * G) {
        if (base != wslot * G) {
            d = base + g; valid = d < D;
            gp = (valid && l < K) ? gprev[(size_t)d * K + l] : (l < K ? 1.0 : 0.0);
            start = valid ? c.doc_ptr[d] : 0;
            W = valid ? (int)(c.doc_ptr[d + 1] - start) : 0;
        }
        const int2* __restrict__ tcd = c.tc + start;
        const double Sp = group_sum<L>(gp);
        lds_wave_sync();
        if (l < KP) myT[l] = (l < K) ? dev_div(gp, Sp) : 0.0;      // (the bits of gp / Sp: both are normal numbers)
        lds_wave_sync();
        double tv[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) tv[k] = myT[k];
        int nchmax = (W + L - 1) / L;
        if (G >= 2) nchmax = max(nchmax, __shfl_xor(nchmax, 32, MMM_WAVE));
        if (G >= 4) nchmax = max(nchmax, __shfl_xor(nchmax, 16, MMM_WAVE));
        nchmax = __builtin_amdgcn_readfirstlane(nchmax);
        for (int j = 0; j < nchmax; ++j) {
            const int w = j * L + l;
            const bool act = w < W;
            const int2 t = act ? tcd[w] : make_int2(0, 0);
            const double* bc = sBeta + (size_t)t.x * KP;
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int k = 0; k + 1 < KP; k += 2) { p0 = fma(tv[k], bc[k], p0); p1 = fma(tv[k + 1], bc[k + 1], p1); }
            if (KP & 1) p0 = fma(tv[KP - 1], bc[KP - 1], p0);
            const double p = act ? p0 + p1 : 1.0;
            acc = fma((double)t.y, dev_log_tab(p, sLog), acc);
        }
    }
    MMM_RSTAMP(lb == 0 && tid == 0, 22);       // sweep done
    acc = wave_sum(acc);
    if (lane == 0) s_w[wid] = acc;
    __syncthreads();
    MMM_RSTAMP(lb == 0 && tid == 0, 23);       // all waves done
    if (tid == 0) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += s_w[w];
        if (cell) cell_store(cell, v, seq); else llpart2[lb] = v;
    }
}

// grid = ceil(V*K/16) blocks of (16 entries, 64 slab lanes): fixed-order (deterministic) sum of the per-block partials
__device__ void lda_reduce_block(const ReduceArgs& r)
{
    __shared__ double sm[64][17];
    const int stop = r.ctl->stop;        // only the stores depend on it: the partial loads below are issued alongside this load
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int e = blockIdx.x * 16 + tx;
    double acc = 0.0;
    if (e < r.VK) for (int sl = ty; sl < r.nslab; sl += 64) acc += r.partial[(size_t)sl * r.VK + e];
    if (stop) {      // a no-op pass still keeps the mailbox rendezvous of its sequence number (p2p.hip header): element 0, value unused
        if (r.p2p && blockIdx.x == 0 && tx == 0 && ty == 0) p2p_send(r.px, r.p2p_seq, 0, 0.0);
        return;
    }
    sm[ty][tx] = acc;
    __syncthreads();
    if (ty < 8) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) v += sm[ty * 8 + j][tx];
        sm[ty * 8][tx] = v;
    }
    __syncthreads();
    if (ty == 0 && e < r.VK) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) v += sm[j * 8][tx];
        r.stats[e] = v;
        if (r.p2p) p2p_send(r.px, r.p2p_seq, e, v);
    }
    if (blockIdx.x == 0 && ty >= 4 && ty < 8 && r.ll_cells && r.do_ll) {      // wave 1 of block 0: ll numerator of pass t-1 from the ll blocks' cells
        const int lane = (ty * 16 + tx) & 63;
        double v = 0.0;
        v = cells_wait_sum(r.ll_cells, r.n_ll, lane, r.ll_seq, r.ctl);
        v = wave_sum(v);
        if (lane == 0) r.stats[r.VK] = v;
    }
    if (blockIdx.x == 0 && ty == 1 && !r.ll_in_k2 && !r.ll_cells) {       // wave 1 of block 0: ll numerator of pass t-1 (from the E-step's partials)
        double v = 0.0;
        for (int i = tx + 16 * 0; i < r.nslab; i += 16) v += r.llpart[i];
        v = group_sum<16>(v);
        if (tx == 0) { r.stats[r.VK] = v; if (r.p2p) p2p_send(r.px, r.p2p_seq, r.VK, v); }
    }
}

__global__ __launch_bounds__(1024) void k_lda_reduce(ReduceArgs r) { lda_reduce_block(r); }

// the same launch with the ll of pass t-1 riding along: blocks [0, nred) are k_lda_reduce's, blocks [nred, gridDim) evaluate the
// log-likelihood numerators while the reduction -- 60 blocks -- leaves most of the chip idle
template <int KP>
__global__ __launch_bounds__(1024) void k_lda_reduce_ll(ReduceArgs r, LdaDev c, const double* gprev, const double* bprev, double* llpart2, int nred)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if ((int)blockIdx.x < nred) { lda_reduce_block(r); return; }
    if (r.ctl->stop) return;
    constexpr int L = KP <= 15 ? 16 : (KP <= 31 ? 32 : 64);       // K <= KP: the E-step's lane-group width (K = 16 -> KP = 16 -> 32 lanes)
    const int lb = (int)blockIdx.x - nred;
    lda_ll_block<KP, L>(c, gprev, bprev, llpart2, lb, (int)gridDim.x - nred, smem, r.ll_cells ? r.ll_cells + 2 * lb : nullptr, r.ll_seq);
}

// ---- V <= 256, plain LDA, no RCCL in the path: the reduction, the ll sweep AND the M-step in one launch.  A topic's column sum needs the V/16
// reduce blocks of that topic; they hand each other their partial column sums through 16-byte cells in device memory --
// {low half | seq} {high half | seq}, complete when both words carry this launch's sequence number, so no fence and no
// flag (the mailbox format of p2p.hip) -- and then run the M-step of their own 16 entries, in parallel, while the ll blocks
// are still sweeping.  Wave 1 of the first reduce block collects the ll blocks' numerators the same way and runs the pass tail.
// The reduce blocks have the lowest block ids (dispatched first; putting the ll blocks first was 0.4 us slower) and wait only for
// each other and for the ll blocks, which wait for nothing; every wait has an iteration cap (ctl->wait_timeout, reported by the
// next host synchronisation).
struct IldaDesc {
    int I, V, K, SJ;
    int J[kIldaMaxI], joff[kIldaMaxI + 1];     // joff = prefix sums of J
    double eta[kIldaMaxI];
    const int* features;                       // [i*V + v], 0-based feature values
};

// ILDA in the merged launch: the factor arrays of the pass's ring slot and the cells the blocks of a topic use to hand each other
// their partial folds (16 per block: sum(J) <= 16)
struct IldaMerge {
    IldaDesc ds;
    double* ilam; double* iEln; double* ibeta;
    unsigned long long* fcells;
};

struct MergeArgs {
    int V; double eta;
    Ring lambda, Elnbeta, expElnbeta, beta;
    unsigned long long* cells;      // [nred] column-sum cells, then [512] ll cells
    unsigned int seq;               // never reused (a discarded pass must not leave valid-looking cells behind)
    int nred;
    int ll_join;                    // large corpora (the ll blocks loop over their documents): the reduce blocks 1.. take a share of the ll sweep
                                    // once their 16 entries are done -- the launch holds only as many blocks as are resident at once (61 of the
                                    // 256 at K = 10, V = 96 are reduce blocks, busy for ~6 us of a ~200 us sweep at 640k documents)
    int n_ll;                       // ll blocks [nredp, nredp + n_ll)
    int epb, nredp;                 // logical reduce blocks (16 entries) per physical block: 1 or 2; physical reduce blocks = nred / epb
    // pro: the ll blocks also form Elntheta_{t+1} = psi(gamma_{t+1}) - psi(sum) and a = exp(Elntheta_{t+1}) (LDA.jl:78-80) of their documents
    // for the NEXT pass's single-step E-step kernel, after their numerator has left (one step covers the corpus: n_ll x 64 >= D)
    int pro;
    const double* pro_gamma; double* pro_Eln; double* pro_a;
};

// P2P: several GPUs with the mailboxes up -- a reduce block sends its 16 sums to the peers and adds theirs (rank order) before the
// column-sum exchange, the tail wave does the same with the ll numerator: the all-reduce rides inside this launch.
// ILDA (sum(J) <= 16, one GPU): the blocks of a topic exchange their partial FOLDS of the statistics onto the feature values
// (one cell per (feature, value)) instead of one column sum, every block forms the topic's lambda[i][j] from them in block
// order, and writes the effective tables of its own 16 entries; the topic's first block also writes the factor arrays.
template <int KP, bool P2P, bool ILDA>
__global__ __launch_bounds__(1024) void k_lda_reduce_ll_mstep(ReduceArgs r, LdaDev c, const double* gprev, const double* bprev, MergeArgs ms, IldaMerge im)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double sm2[2][64][17];
    const int stop = r.ctl->stop;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * 16 + tx;
    MMM_RSTAMP(blockIdx.x == 1 && tid == 0, 0);                         // reduce block 1, wave 0
    MMM_RSTAMP(blockIdx.x == 0 && tid == 64, 8);                        // tail wave
    MMM_RSTAMP((int)blockIdx.x == ms.nredp && tid == 0, 16);             // first ll block
    if ((int)blockIdx.x >= ms.nredp) {       // ---- ll block: numerator of pass t-1 into its cell
        if (stop) return;
        constexpr int L = KP <= 15 ? 16 : (KP <= 31 ? 32 : 64);
        const int lb = (int)blockIdx.x - ms.nredp, n_ll = ms.n_ll;
        MMM_LLSTAMP(0, lb);
        // ms.pro: the block also runs the NEXT pass's prologue for its documents (the four of each wave, 16 lanes per document as
        // k_lda_estep<., 16, ...> has them): gamma_{t+1} is requested before the sweep and used after the block's numerator has left --
        // the pass tail (wave 1 of block 0, the end of this launch's critical path) does not wait a cycle longer for it
        double gnx = 0.0;
        int pd = 0;
        if constexpr (KP <= 12) {
            if (ms.pro) {
                const int lane = tid & 63, l = lane & 15;
                pd = ((tid >> 6) * n_ll + lb) * 4 + (lane >> 4);          // lda_ll_block's wave slots
                gnx = (pd < c.D && l < c.K) ? ms.pro_gamma[(size_t)pd * c.K + l] : (l < c.K ? 1.0 : 0.0);
            }
        }
        lda_ll_block<KP, L>(c, gprev, bprev, nullptr, lb, n_ll + (ms.ll_join ? ms.nredp - 1 : 0), smem, ms.cells + 2 * (ms.nred + lb), ms.seq);
        MMM_RSTAMP((int)blockIdx.x == ms.nredp && tid == 0, 17);
        MMM_LLSTAMP(1, lb);
        if constexpr (KP <= 12) {
            if (ms.pro) {          // Elntheta_{t+1}, exp(Elntheta_{t+1}) (LDA.jl:78-80): the operations of the E-step kernel's prologue
                const int lane = tid & 63, g = lane >> 4, l = lane & 15, K = c.K;
                const double S = group_sum<16>(gnx);
                const double ps = dev_digamma_pos(l < K ? gnx : S);        // lane K of the group holds psi(S)
                const double psS = __shfl(ps, g * 16 + K, MMM_WAVE);
                const double el = ps - psS;
                if (pd < c.D && l < K) { ms.pro_Eln[(size_t)pd * K + l] = el; ms.pro_a[(size_t)pd * K + l] = ar_exp(el); }
            }
        }
        return;
    }
    // ---- reduce block: epb (1 or 2) groups of 16 entries of the statistics (each as lda_reduce_block).  With two groups the block is two of
    //      the `nred` logical reduce blocks, 2 rbp and 2 rbp + 1: every thread sums both groups' partials, and rows 0 and 1 of wave 0 (ty = 0 / 1)
    //      finish one group each in lock step -- same sums, same cells, half the CUs (the others go to the ll sweep, which bounds the launch)
    const int rbp = (int)blockIdx.x, epb = ms.epb;      // physical reduce block
    const int rb = rbp * epb + (ty < epb ? ty : 0);     // the logical reduce block of this thread's row in the final stage
    const int e = rb * 16 + tx;
    double acc = 0.0, acc1 = 0.0;
    {
        const int e0 = rbp * epb * 16 + tx;
        // (one loop for both groups: all of a thread's loads leave together -- as two loops the second group's waited for the first's:
        // partial sums in hand after 3.1 instead of 1.9 us; with the mailbox exchange on this chain that would be the launch's critical path)
        if (epb == 2) {
            for (int sl = ty; sl < r.nslab; sl += 64) { acc += r.partial[(size_t)sl * r.VK + e0]; acc1 += r.partial[(size_t)sl * r.VK + e0 + 16]; }
        } else {
            for (int sl = ty; sl < r.nslab; sl += 64) acc += r.partial[(size_t)sl * r.VK + e0];
        }
    }
    if (stop) {      // a no-op pass still keeps the mailbox rendezvous of its sequence number (p2p.hip header): element 0, value unused
        if (P2P && rbp == 0 && tid == 0) { p2p_send(r.px, r.p2p_seq, 0, 0.0); (void)p2p_recv_sum(r.px, r.p2p_seq, 0, 0.0); }
        return;
    }
    MMM_RSTAMP(blockIdx.x == 1 && tid == 0, 1);                         // partial loads done
    sm2[0][ty][tx] = acc; sm2[1][ty][tx] = acc1;
    __syncthreads();
    if (ty < 8) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double v = 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) v += sm2[h][ty * 8 + j][tx];
            sm2[h][ty * 8][tx] = v;
        }
    }
    __syncthreads();
    double (*sm)[17] = sm2[ty < epb ? ty : 0];
    MMM_RSTAMP(blockIdx.x == 1 && tid == 0, 2);                         // tree done
    if (ty < epb) {                          // lanes 0..15 (and 16..31: the block's second group) of wave 0
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) v += sm[j * 8][tx];
        if (P2P) { p2p_send(r.px, r.p2p_seq, e, v); v = p2p_recv_sum(r.px, r.p2p_seq, e, v); }
        r.stats[e] = v;
        // ---- M-step of these 16 entries (LDA.jl:96-112): column sum = the topic's block sums in block order.  Rows are padded
        //      to a multiple of 16 (Vp): a block never straddles two topics; pad entries carry zeros and are not written back
        const int V = ms.V, Vp = (V + 15) & ~15, nb = Vp / 16, k = e / Vp, vv = e - k * Vp, slot = r.t % 3;
        const bool real = vv < V;
        if (ILDA) {
            const IldaDesc& ds = im.ds;
            const int K = ds.K, SJ = ds.SJ;
            // (feature, value) of lane q = tx; partial fold of this block's 16 entries onto it (ILDA.jl:107-126)
            int qi = 0, qj = 0;
            double mine = 0.0;
            for (int i = 0; i < ds.I; ++i) {
                const int fi = real ? ds.features[(size_t)i * V + vv] : -1;
                for (int j = 0; j < ds.J[i]; ++j) {
                    const double pq = group_sum<16>(fi == j ? v : 0.0);
                    if (tx == ds.joff[i] + j) { mine = pq; qi = i; qj = j; }
                }
            }
            if (tx < SJ) cell_store(im.fcells + 2 * ((size_t)rb * 16 + tx), mine, ms.seq);
            double lam = 0.0;
            if (tx < SJ) {
                lam = ds.eta[qi];
                for (int b = 0; b < nb; ++b) lam += cell_wait(im.fcells + 2 * ((size_t)(k * nb + b) * 16 + tx), ms.seq, r.ctl);
            }
            double cs = 0.0;
            for (int i = 0; i < ds.I; ++i) {
                const double ci = group_sum<16>((tx < SJ && qi == i) ? lam : 0.0);
                if (tx < SJ && qi == i) cs = ci;
            }
            double el = 0.0, bq = 1.0;
            if (tx < SJ) {
                el = dev_digamma_pos(lam) - dev_digamma_pos(cs); bq = lam / cs;
                if (vv == tx) {         // the topic's first block (its lane 0 sits on entry 0 of the row) keeps the model arrays (ILDA.jl:6-9 layout)
                    const size_t o = (size_t)K * ds.joff[qi] + (size_t)ds.J[qi] * k + qj;
                    im.ilam[o] = lam; im.iEln[o] = el; im.ibeta[o] = bq;
                }
            }
            lds_wave_sync();
            sm[1][tx] = el; sm[2][tx] = bq;          // (the reduction tree above is done with sm)
            lds_wave_sync();
            if (real) {
                double ee = 0.0, bb = 1.0;
                for (int i = 0; i < ds.I; ++i) { const int q = ds.joff[i] + ds.features[(size_t)i * V + vv]; ee += sm[1][q]; bb *= sm[2][q]; }
                const size_t o = (size_t)k * V + vv;
                ms.Elnbeta.s[slot][o] = ee; ms.expElnbeta.s[slot][o] = exp(ee); ms.beta.s[slot][o] = bb;
            }
        } else {
        const double lam = real ? ms.eta + v : 0.0;
        const double part = group_sum<16>(lam);
        if (tx == 0) cell_store(ms.cells + 2 * rb, part, ms.seq);
        const double got = (tx < nb) ? cell_wait(ms.cells + 2 * (k * nb + tx), ms.seq, r.ctl) : 0.0;
        double cs = 0.0;
        for (int j = 0; j < nb; ++j) cs += __shfl(got, j, 16);
        MMM_RSTAMP(blockIdx.x == 1 && tid == 0, 3);                     // column sum in hand
        const double el = dev_digamma_pos(real ? lam : 1.0) - dev_digamma_pos(cs);
        const size_t o = (size_t)k * V + vv;
        if (real) { ms.lambda.s[slot][o] = lam; ms.Elnbeta.s[slot][o] = el; ms.expElnbeta.s[slot][o] = exp(el); ms.beta.s[slot][o] = lam / cs; }
        MMM_RSTAMP(blockIdx.x == 1 && tid == 0, 4);                     // M-step stores done
        }
    }
    if (ms.ll_join && rbp > 0) {             // (uniform per block; block 0 keeps the pass tail)
        constexpr int L = KP <= 15 ? 16 : (KP <= 31 ? 32 : 64);
        const int n_ll = ms.n_ll, lb = n_ll + rbp - 1;
        lda_ll_block<KP, L>(c, gprev, bprev, nullptr, lb, n_ll + ms.nredp - 1, smem, ms.cells + 2 * (ms.nred + lb), ms.seq);
        return;
    }
    if (rbp == 0 && ty >= 4 && ty < 8) {     // wave 1 of block 0: ll numerator of pass t-1, stopping rule, pass counter
        const int lane = tid & 63, n_ll = ms.n_ll + (ms.ll_join ? ms.nredp - 1 : 0);
        // what the tail needs from memory is fetched before the wait, not after it (lda_pass_tail's dependent loads)
        const int n = r.ctl->n_hist;
        const double prev = (r.do_ll && n > 0) ? r.ll_hist[n - 1] : 0.0;
        double v = 0.0;
        if (r.do_ll) {
            MMM_RSTAMP(lane == 0, 9);
            v = cells_wait_sum(ms.cells + 2 * ms.nred, n_ll, lane, ms.seq, r.ctl);
            MMM_RSTAMP(lane == 0, 10);
            v = wave_sum(v);
            if (P2P && lane == 0) { p2p_send(r.px, r.p2p_seq, r.VK, v); v = p2p_recv_sum(r.px, r.p2p_seq, r.VK, v); }
        }
        if (lane == 0) {
            int halt = 0;
            if (r.do_ll) {
                r.stats[r.VK] = v;
                const double ll = v / r.Nglobal;
                r.ll_hist[n] = ll;
                r.ctl->n_hist = n + 1;
                if (n + 1 - r.conv_base > 10 && fabs(prev - ll) / fabs(ll) < r.tol) { halt = 1; r.ctl->stop = 1; r.ctl->stop_iter = r.t - 1; }      // common.jl:53-56
            }
            if (!halt) r.ctl->t = r.t;
            r.ctl->ticket = 0;
        }
        MMM_RSTAMP(lane == 0, 11);
    }
}

// M-step of pass t from the (all-reduced) statistics, one wave per topic (no inter-block dependency: Elnbeta_k needs
// only the column sum of topic k): lambda = eta + sums, Elnbeta, exp table, beta (LDA.jl:96-112); block 0 then finalises
// ll_{t-1}, the stopping rule and the pass counter.
// Blocks of two waves: with the mailbox exchange folded in, both waves receive (V <= 128 entries in ONE polling round);
// wave 0 alone then runs the topic's M-step.
// P2P = false: the build without the mailbox code (its polling arrays live in scratch memory; a single-GPU launch carries none).
template <bool P2P>
__global__ __launch_bounds__(128) void k_lda_mstep(ReduceArgs r, int V, double eta, Ring lambda, Ring Elnbeta, Ring expElnbeta, Ring beta)
{
    const int stop = r.ctl->stop;
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, c = r.t % 3;
    if (k == (int)gridDim.x - 1) {      // the extra block: pass tail, concurrent with the topic blocks (its loads are a dependent chain)
        if (!stop && tid < 64) lda_tail_block<P2P>(r, lane);
        return;
    }
    double* sums = r.stats + (size_t)k * V;
    if (P2P && r.p2p) {             // all-reduce folded in: own statistics + the peers', summed in rank order, written back for the passes below
        if (!stop) for (int v = tid; v < V; v += 128) sums[v] = p2p_recv_sum(r.px, r.p2p_seq, k * V + v, sums[v]);
        else if (k == 0 && tid == 0) (void)p2p_recv_sum(r.px, r.p2p_seq, 0, 0.0);      // no-op pass: the rendezvous of lda_reduce_block's dummy send
        __syncthreads();
    }
    // both waves form the column sum (same loads, same order: same bits, and no barrier); each then takes every other 64 entries
    double part = 0.0;
    for (int v = lane; v < V; v += 64) part += eta + sums[v];
    if (stop) return;
    const double cs = wave_sum(part);
    const double psi = dev_digamma_pos(cs);
    for (int v = tid; v < V; v += 128) {
        const double l = eta + sums[v];
        const double el = dev_digamma_pos(l) - psi;
        const size_t e = (size_t)k * V + v;
        lambda.s[c][e] = l; Elnbeta.s[c][e] = el; expElnbeta.s[c][e] = exp(el); beta.s[c][e] = l / cs;
    }
}

// k_lda_mstep for wide vocabularies: 512 threads per topic instead of one wave (V in the thousands), no folded exchange
__global__ __launch_bounds__(512) void k_lda_mstep_wide(ReduceArgs r, int V, double eta, Ring lambda, Ring Elnbeta, Ring expElnbeta, Ring beta)
{
    __shared__ double sh[16];
    const int stop = r.ctl->stop;
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, c = r.t % 3;
    if (k == (int)gridDim.x - 1) {
        if (!stop && tid < 64) lda_tail_block<false>(r, lane);
        return;
    }
    if (stop) return;
    const double* sums = r.stats + (size_t)k * V;
    double part = 0.0;
    for (int v = tid; v < V; v += 512) part += eta + sums[v];
    part = wave_sum(part);
    if (lane == 0) sh[wid] = part;
    __syncthreads();
    double cs = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) cs += sh[w];
    const double psi = dev_digamma_pos(cs);
    for (int v = tid; v < V; v += 512) {
        const double l = eta + sums[v];
        const double el = dev_digamma_pos(l) - psi;
        const size_t e = (size_t)k * V + v;
        lambda.s[c][e] = l; Elnbeta.s[c][e] = el; expElnbeta.s[c][e] = exp(el); beta.s[c][e] = l / cs;
    }
}

// ---- ILDA (src/ILDA.jl): LDA whose topic-term distribution factorises over I features of the term, beta_kv = prod_i
// beta[i][f_vi, k].  The E-step, ll and ELBO document kernels run unchanged on EFFECTIVE V x K tables (Elnbeta_eff[v,k] =
// sum_i Elnbeta[i][f_vi, k], exp of it, beta_eff = prod_i beta[i][f_vi, k]); only the topic M-step differs: the V x K
// statistics are folded onto the feature values.  Model layout: lambda[i] is J_i x K column-major at K * sum_{q<i} J_q.

// mode 0: lambda = eta + folded sums (update_λ!, ILDA.jl:107-126); 1: from the stored lambda (update_Elnβ!/update_β!,
// :97-104,128-130); 2: effective tables only, from the stored Elnbeta / beta (after an upload).  One wave per topic.
__global__ __launch_bounds__(64 * kIldaMaxI) void k_ilda_mstep(IldaDesc ds, int mode, const double* sums, double* ilam, double* iEln, double* ibeta,
                                                   double* Eeff, double* expEeff, double* beff, const int* stop, int write_beta_only,
                                                   ReduceArgs tail, int with_tail)
{
    __shared__ double sE[kIldaMaxSJ], sB[kIldaMaxSJ];
    if (stop && *stop) return;
    // one wave per feature (launch: 64 * I threads): the features' folds and digamma chains run side by side
    const int k = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nwv = blockDim.x >> 6, V = ds.V, K = ds.K;
    if (with_tail && k == K) {           // extra block of the fused pass: ll_{t-1}, stopping rule, pass counter (as k_lda_mstep)
        if (wid == 0) lda_tail_block<false>(tail, lane);      // (the ILDA exchange is never folded)
        return;
    }
    // V <= 256 (the 96 SNV contexts): the topic's statistics are fetched once, four per lane, and every masked sum below runs
    // out of registers (same lane assignment and order as the general loop, so the same bits)
    const bool in_regs = mode == 0 && V <= 256;
    double sv[4] = {0.0, 0.0, 0.0, 0.0};
    if (in_regs) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int v = lane + 64 * q; if (v < V) sv[q] = sums[(size_t)k * V + v]; }
    }
    for (int i = wid; i < ds.I; i += nwv) {
        const int Ji = ds.J[i];
        const size_t base = (size_t)K * ds.joff[i] + (size_t)Ji * k;
        const int* f = ds.features + (size_t)i * V;
        int fv[4] = {-1, -1, -1, -1};
        if (in_regs) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int v = lane + 64 * q; if (v < V) fv[q] = f[v]; }
        }
        double part = 0.0;
        for (int j0 = 0; j0 < Ji; j0 += 64) {
            const int j = j0 + lane;
            double l = 0.0;
            if (mode == 0) {
                // fold the topic's V statistics onto this feature's values: all 64 lanes walk the terms, one butterfly sum per
                // value (a lane-per-value loop over V global loads is a 60 us dependent chain)
                for (int jj = j0; jj < min(Ji, j0 + 64); ++jj) {
                    double t = 0.0;
                    if (in_regs) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) t += (fv[q] == jj) ? sv[q] : 0.0;
                    } else
                        for (int v = lane; v < V; v += 64) t += (f[v] == jj) ? sums[(size_t)k * V + v] : 0.0;
                    t = wave_sum(t);
                    if (lane == jj - j0) l = ds.eta[i] + t;
                }
                if (j < Ji) ilam[base + j] = l;
            } else if (j < Ji) l = ilam[base + j];
            part += l;
        }
        const double cs = wave_sum(part);
        const double psi = dev_digamma_pos(cs);
        for (int j0 = 0; j0 < Ji; j0 += 64) {
            const int j = j0 + lane;
            if (j < Ji) {
                double el, b;
                if (mode == 2) { el = iEln[base + j]; b = ibeta[base + j]; }
                else {
                    const double l = ilam[base + j];
                    el = dev_digamma_pos(l) - psi; b = l / cs;
                    if (!write_beta_only) iEln[base + j] = el;
                    ibeta[base + j] = b;
                }
                sE[ds.joff[i] + j] = el; sB[ds.joff[i] + j] = b;
            }
        }
    }
    __syncthreads();
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        double e = 0.0, b = 1.0;
        for (int i = 0; i < ds.I; ++i) { const int j = ds.features[(size_t)i * V + v]; e += sE[ds.joff[i] + j]; b *= sB[ds.joff[i] + j]; }
        const size_t o = (size_t)k * V + v;
        if (!write_beta_only) { Eeff[o] = e; expEeff[o] = exp(e); }
        beff[o] = b;
    }
}

// Frozen-topic passes (transform / fit_heldout, LDA.jl:233-295): the E-step kernel runs with fixed tables and evaluates
// the ll of the SAME pass (theta_t and beta are both known); this kernel sums the per-block numerators (phase & 1), and
// (phase & 2) records ll_t, applies the stopping rule (LDA.jl:252 / :285) and advances the pass counter.
__global__ __launch_bounds__(64) void k_lda_infer_tail(ReduceArgs r, int phase)
{
    if (r.ctl->stop) return;
    const int lane = threadIdx.x;
    if (phase & 1) {
        double v = 0.0;
        for (int i = lane; i < r.nslab; i += 64) v += r.llpart[i];
        v = wave_sum(v);
        if (lane == 0) r.stats[r.VK] = v;
    }
    if ((phase & 2) && lane == 0) {
        const int n = r.ctl->n_hist;
        const double ll = r.stats[r.VK] / r.Nglobal;
        r.ll_hist[n] = ll;
        r.ctl->n_hist = n + 1;
        if (n + 1 - r.conv_base > 10) {
            const double prev = r.ll_hist[n - 1];
            if (fabs(prev - ll) / fabs(ll) < r.tol) { r.ctl->stop = 1; r.ctl->stop_iter = r.t; }
        }
        r.ctl->t = r.t;           // the state of the stopping pass is kept (its ll is not lagged)
    }
}

__global__ void k_lda_tail_only(ReduceArgs r) { if (!r.ctl->stop) lda_pass_tail(r); }
