/*
 * mmm_twin.c -- CPU ORACLE, order-matched variant (test infrastructure only; see mmm_oracle.h for the contract).
 *
 * mmm_oracle.c restates fit!(::MMCTM / ::IMMCTM) with every sum in index order.  This file restates the SAME algorithm
 * (same reference lines, cited per function) with every floating-point sum associated the way the gfx950 kernels of
 * multimodalmusig.jl_amd/csrc/ctm.hip associate it -- balanced trees over the lanes of a document group, four interleaved
 * chains in the matrix-vector product, per-wave slabs / per-block partials / 64-8-8 folds across documents -- and with the
 * one restatement of exp / log / digamma both sides compile (csrc/mmm_arith.h).  Neither order is "the reference's": Julia's
 * own `sum` is pairwise above 16 elements and its BLAS picks an order per CPU.  Both variants are pinned to the same known-answer
 * tests (tests/test_oracle_kats.py) and to each other (tests/test_twin_cpu.py: one pass from identical state <= 1e-12).
 *
 * Why it exists: NLopt's LD_MMA stops on discontinuous tests; only an evaluation that is bit-identical on both sides takes
 * the same stopping decisions for every document.  With this variant the device's whole-fit trajectory can be compared with
 * a CPU restatement bit for bit (tests/test_ctm_gpu.py, tests/test_brca_gpu.py), and the fork between the two CPU variants
 * measures how far two equally faithful evaluations of the reference drift apart.
 *
 * The sums across documents depend on the launch geometry of the device handle (documents per wave, waves per block, grid
 * sizes); the caller copies it from mmm_ctm_geometry() into orc_ctm.{L, grid_e, waves_e, grid_m}.
 */
#include "mmm_oracle.h"
#include "../multimodalmusig.jl_amd/csrc/mmm_arith.h"
#include "../multimodalmusig.jl_amd/csrc/mmm_exptab.h"
#include "../multimodalmusig.jl_amd/csrc/mmm_logtab.h"

/* the function tables of the LD_MMA objectives (ar_exp_tab / ar_log_tab): the numbers the solve kernels stage into LDS */
static const double tw_exptab[MMM_EXPTAB_N] = { MMM_EXPTAB_VALUES };
static const double tw_logtab[MMM_LOGTAB_N] = { MMM_LOGTAB_VALUES };

/* debug hook (round 5): the smallest and largest argument the table-driven exp / log have been called with by the objectives since the last
 * reset -- tests/test_twin_cpu.py sweeps both functions against mpmath over exactly the range the solves of configs 3-5 reach
 * (tests/golden/table_argument_ranges.json).  Finite arguments only.  Not thread-safe: the restatement is sequential. */
static double tw_arg_range[4] = { 1e300, -1e300, 1e300, -1e300 };      /* exp min, exp max, log min, log max */
static double tw_exp_rec(double x)
{
    if (x == x && x - x == 0.0) { if (x < tw_arg_range[0]) tw_arg_range[0] = x; if (x > tw_arg_range[1]) tw_arg_range[1] = x; }
    return ar_exp_tab(x, tw_exptab);
}
static double tw_log_rec(double x)
{
    if (x == x && x - x == 0.0) { if (x < tw_arg_range[2]) tw_arg_range[2] = x; if (x > tw_arg_range[3]) tw_arg_range[3] = x; }
    return ar_log_tab(x, tw_logtab);
}
void orc_twin_arg_ranges(double out[4], int reset)
{
    for (int i = 0; i < 4; ++i) out[i] = tw_arg_range[i];
    if (reset) { tw_arg_range[0] = 1e300; tw_arg_range[1] = -1e300; tw_arg_range[2] = 1e300; tw_arg_range[3] = -1e300; }
}

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- lane collectives ------------------------------------------------------------------------------------------- */
/* group_sum<L> of dev_math.h: xor-butterfly over offsets 1, 2, 4, 8 (DPP quad_perm / row_half_mirror / row_mirror, which on
 * values already equal inside the mirrored halves are the xor-4 / xor-8 exchanges), then 16, 32.  Every stage adds two
 * values commutatively, so all lanes end with the same bits: a balanced tree over adjacent pairs.  v has L entries. */
static double tw_group_sum(const double* v, int L)
{
    double t[64];
    memcpy(t, v, sizeof(double) * (size_t)L);
    if (L & (L - 1)) {
        /* packed groups (packed_sum<LP> of ctm.hip, L = sum K lanes per document): lane l+off is folded onto lane l for
         * off = 8, 4, 2, 1 (lanes without a partner add a spare lane's 0), lane 0 holds the total */
        for (int off = 8; off >= 1; off >>= 1) {
            if (off >= L) continue;
            for (int l = 0; l < off && l + off < L; ++l) t[l] += t[l + off];
        }
        return t[0];
    }
    for (int n = L; n > 1; n >>= 1)
        for (int i = 0; i < n / 2; ++i) t[i] = t[2 * i] + t[2 * i + 1];
    return t[0];
}

/* wave_sum of dev_math.h: xor-butterfly over offsets 32, 16, 8, 4, 2, 1 on 64 lanes */
static double tw_wave_sum(const double* v)
{
    double t[64], u[64];
    memcpy(t, v, sizeof t);
    for (int off = 32; off > 0; off >>= 1) {
        for (int i = 0; i < 64; ++i) u[i] = t[i] + t[i ^ off];
        memcpy(t, u, sizeof t);
    }
    return t[0];
}

/* k_reduce_partials: part[nslab][n] -> out[n]; thread row ty sums slabs ty, ty+64, ...; then 8 folds of 8; then a fold of 8 */
static void tw_reduce_partials(const double* part, int nslab, int n, double* out)
{
    for (int e = 0; e < n; ++e) {
        double sm[64];
        for (int ty = 0; ty < 64; ++ty) {
            double acc = 0.0;
            for (int sl = ty; sl < nslab; sl += 64) acc += part[(size_t)sl * n + e];
            sm[ty] = acc;
        }
        double f[8];
        for (int ty = 0; ty < 8; ++ty) { double v = 0.0; for (int j = 0; j < 8; ++j) v += sm[ty * 8 + j]; f[ty] = v; }
        double v = 0.0;
        for (int j = 0; j < 8; ++j) v += f[j];
        out[e] = v;
    }
}

/* ---- geometry helpers --------------------------------------------------------------------------------------------- */
static int tw_koff(const orc_ctm* m, int mod) { int o = 0; for (int i = 0; i < mod; ++i) o += m->K[i]; return o; }
static int tw_goff(const orc_ctm* m, int mod) { int o = 0; for (int i = 0; i < mod; ++i) o += m->K[i] * m->V[i]; return o; }   /* effective [k][v] tables */
static int tw_aoff(const orc_ctm* m, int mod) { int o = 0; for (int i = 0; i < mod; ++i) o += m->n_feat[i]; return o; }
static int tw_SJ(const orc_ctm* m, int mod) { int o = tw_aoff(m, mod), s = 0; for (int i = 0; i < m->n_feat[mod]; ++i) s += m->J[o + i]; return s; }
static size_t tw_mgoff(const orc_ctm* m, int mod) { size_t o = 0; for (int i = 0; i < mod; ++i) o += (size_t)m->K[i] * (m->n_feat ? tw_SJ(m, i) : m->V[i]); return o; }
static size_t tw_foff(const orc_ctm* m, int mod) { size_t o = 0; for (int i = 0; i < mod; ++i) o += (size_t)m->n_feat[i] * m->V[i]; return o; }
static size_t tw_toff(const orc_ctm* m, int mod) {
    size_t o = 0;
    for (int i = 0; i < mod; ++i) o += (size_t)(m->doc_ptr[(size_t)i * (m->D + 1) + m->D] - m->doc_ptr[(size_t)i * (m->D + 1)]) * m->K[i];
    return o;
}
static int tw_GT(const orc_ctm* m) { return tw_goff(m, m->M); }

static double tw_digamma(double x)
{
    if (x > 0.0 && x < 1e40) return ar_digamma_pos(x);
    return orc_digamma(x);          /* never taken for Dirichlet parameters */
}

/* ---- objectives in NLopt's minimisation form, lane layout (common.jl:11-36 negated) -------------------------------- */
typedef struct {
    int n, L;
    int cpl;                  /* > 1: k_ctm_solve_cpl -- n / cpl lanes per document, cpl coordinates per lane (L = n) */
    const double *other;      /* nu: lambda ; lambda: nu */
    const double *c;          /* Ndivzeta per coordinate */
    const double *sumth;      /* lambda objective */
    const double *mu, *invSigma;
} tw_obj;

/* sum over a document's coordinates as the solve kernel in use associates it: the lane butterfly over one coordinate per lane
 * (mma_group), or (mma_cpl) every lane adds its cpl coordinates in index order from 0 and the lanes' sums go through the butterfly */
static double tw_sum(const tw_obj* o, const double* t)
{
    if (o->cpl > 1) {
        double part[64];
        const int act = o->n / o->cpl;          /* lanes that hold coordinates; the document's remaining lanes contribute 0 */
        int lpd = 1;
        while (lpd < act) lpd <<= 1;
        for (int l = 0; l < lpd; ++l) {
            double s = 0.0;
            if (l < act) for (int q = 0; q < o->cpl; ++q) s += t[l * o->cpl + q];
            part[l] = s;
        }
        return tw_group_sum(part, lpd);
    }
    return tw_group_sum(t, o->L);
}

/* nu: f = 1/2 sum nu_i S_ii + sum c_i exp(lambda_i + nu_i/2) - 1/2 sum log nu_i   (NuObj::eval) */
static double tw_nu_eval(const tw_obj* o, const double* x, double* g)
{
    double t[64];
    const int n = o->n;
    for (int l = 0; l < o->L; ++l) {
        if (l >= n) { t[l] = 0.0; g[l] = 0.0; continue; }
        const double Sll = o->invSigma[(size_t)l * n + l], c = o->c[l];
        const double E = tw_exp_rec(o->other[l] + 0.5 * x[l]);
        g[l] = 0.5 * Sll + 0.5 * c * E - 1.0 / (2.0 * x[l]);
        t[l] = 0.5 * x[l] * Sll + c * E - 0.5 * tw_log_rec(x[l]);
    }
    return tw_sum(o, t);
}

/* lambda: f = 1/2 (x-mu)' S (x-mu) - x . sumtheta + sum c_i exp(x_i + nu_i/2)   (LamObj::eval) */
static double tw_lam_eval(const tw_obj* o, const double* x, double* g)
{
    double t[64], diff[64];
    const int n = o->n;
    for (int l = 0; l < n; ++l) diff[l] = x[l] - o->mu[l];
    for (int l = 0; l < o->L; ++l) {
        if (l >= n) { t[l] = 0.0; g[l] = 0.0; continue; }
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int j = 0;
        for (; j + 3 < n; j += 4) {
            s0 = fma(o->invSigma[(size_t)j * n + l], diff[j], s0);
            s1 = fma(o->invSigma[(size_t)(j + 1) * n + l], diff[j + 1], s1);
            s2 = fma(o->invSigma[(size_t)(j + 2) * n + l], diff[j + 2], s2);
            s3 = fma(o->invSigma[(size_t)(j + 3) * n + l], diff[j + 3], s3);
        }
        for (; j < n; ++j) s0 = fma(o->invSigma[(size_t)j * n + l], diff[j], s0);
        const double Sd = (s0 + s1) + (s2 + s3);
        const double c = o->c[l], sumth = o->sumth[l];
        const double E = tw_exp_rec(x[l] + 0.5 * o->other[l]);
        g[l] = Sd - sumth + c * E;
        t[l] = 0.5 * diff[l] * Sd - x[l] * sumth + c * E;
    }
    return tw_sum(o, t);
}

typedef double (*tw_eval_fn)(const tw_obj*, const double*, double*);

/* NLopt LD_MMA with zero constraints, one document group (mma_group<L> of ctm.hip; algorithm statement: mmm_oracle.c
 * orc_mma_minimize).  x: n coordinates, in/out.  Returns the number of objective evaluations (negative: cap hit). */
static int tw_mma(const tw_obj* o, tw_eval_fn eval, double* x, int has_lb, double lb, double xtol_rel, double xtol_abs,
                  int xtol_rule, int max_eval)
{
    const int n = o->n, L = o->L;
    double sigma[64], grad[64], gcur[64], xc[64], xcur[64], xprev[64], xprevprev[64], gl[64], wl[64], tmp[64], tmp2[64];
    double rho = 1.0;
    for (int l = 0; l < L; ++l) { sigma[l] = 1.0; if (l >= n) x[l] = 0.0; }
    double fbest = eval(o, x, grad);
    for (int l = 0; l < L; ++l) { xcur[l] = x[l]; xprev[l] = x[l]; xprevprev[l] = x[l]; }
    int k = 1, nev = 1;
    const int cap = max_eval > 0 ? max_eval : 2000;
    for (;;) {
        for (int l = 0; l < L; ++l) {
            if (l >= n) { xc[l] = x[l]; gl[l] = 0.0; wl[l] = 0.0; continue; }
            /* the device's form of NLopt's step (ctm_estep.cuh mma_group): dx = (u/v) / (-1 - sqrt|1 - (u/(v sigma))^2|) multiplied through by v,
             * with v^2 - g^2 sigma^2 = rho (|g| sigma + rho/4) -- one quotient, one root, no cancellation */
            const double sigma2 = sigma[l] * sigma[l];
            const double ags = fabs(grad[l]) * sigma[l];
            const double v = ags + 0.5 * rho;
            const double gs2 = grad[l] * sigma2;
            double dx = -gs2 / (v + sqrt(rho * (ags + 0.25 * rho)));
            double c = x[l] + dx;
            if (has_lb && c < lb) c = lb;
            if (c > x[l] + 0.9 * sigma[l]) c = x[l] + 0.9 * sigma[l]; else if (c < x[l] - 0.9 * sigma[l]) c = x[l] - 0.9 * sigma[l];
            xc[l] = c;
            dx = c - x[l];
            const double dx2 = dx * dx;
            const double denominv = 1.0 / (sigma2 - dx2);
            gl[l] = (fma(v, dx, gs2) * dx) * denominv;      /* = (g sigma^2 dx + v dx^2) / (sigma^2 - dx^2), with the step's own g sigma^2 */
            wl[l] = dx2 * denominv;                         /* (the 1/2 is applied to the sum: the same bits) */
        }
        const double gval = fbest + tw_sum(o, gl);
        const double wval = 0.5 * tw_sum(o, wl);
        const double fcur = eval(o, xc, gcur);
        ++nev;
        memcpy(xcur, xc, sizeof(double) * (size_t)L);
        int inner_done = gval >= fcur;
        if (fcur < fbest) { fbest = fcur; memcpy(x, xc, sizeof(double) * (size_t)L); memcpy(grad, gcur, sizeof(double) * (size_t)L); }
        if (nev >= cap) return -nev;
        if (!inner_done) {
            if (fcur > gval) { const double a = 10.0 * rho, b = 1.1 * (rho + (fcur - gval) / wval); rho = fmin(a, b); }
            continue;
        }
        /* outer iteration finished: NLopt's x-tolerance test on (xcur, xprev) */
        int stop;
        if (xtol_rule == 0) {
            for (int l = 0; l < L; ++l) { tmp[l] = l < n ? fabs(xcur[l] - xprev[l]) : 0.0; tmp2[l] = l < n ? fabs(xcur[l]) : 0.0; }
            const double dn = tw_sum(o, tmp), xn = tw_sum(o, tmp2);
            int all_abs = 1;
            for (int l = 0; l < n; ++l) if (!(tmp[l] < xtol_abs)) all_abs = 0;
            stop = (dn < xtol_rel * xn) || all_abs;
        } else {
            stop = 1;
            for (int l = 0; l < n; ++l) {
                const double ad = fabs(xcur[l] - xprev[l]);
                const int ok = isinf(xprev[l]) ? 0 : (ad < xtol_abs || ad < xtol_rel * (fabs(xcur[l]) + fabs(xprev[l])) * 0.5 || (xtol_rel > 0 && xcur[l] == xprev[l]));
                if (!ok) stop = 0;
            }
        }
        if (stop) return nev;
        rho = fmax(0.1 * rho, 1e-5);
        if (k > 1)
            for (int l = 0; l < n; ++l) {
                const double sgn = (xcur[l] - xprev[l]) * (xprev[l] - xprevprev[l]);
                sigma[l] *= (sgn < 0 ? 0.7 : (sgn > 0 ? 1.2 : 1.0));
            }
        ++k;
        memcpy(xprevprev, xprev, sizeof(double) * (size_t)L);
        memcpy(xprev, xcur, sizeof(double) * (size_t)L);
    }
}

/* ---- update_γ! (statistics -> gamma) / update_Elnϕ! / update_ϕ!: k_ctm_mstep_topics ------------------------------- */
/* MMCTM.jl:214-250; IMMCTM.jl:188-223.  sG: gamma statistics in the effective [m][k][v] layout (NULL: keep gamma).
 * Writes gamma, Elnphi (model layout), expE (effective layout, m->expE) and phi (MMCTM). */
void orc_twin_topics(orc_ctm* m, const double* sG)
{
    for (int mod = 0; mod < m->M; ++mod) {
        const int Km = m->K[mod], Vm = m->V[mod], go = tw_goff(m, mod);
        for (int k = 0; k < Km; ++k) {
            if (!m->n_feat) {
                double* gam = m->gamma + go + (size_t)k * Vm; double* El = m->Elnphi + go + (size_t)k * Vm;
                /* 256 threads: thread tid sums v = tid, tid+256, ...; wave_sum per wave; ((sh0+sh1)+sh2)+sh3 */
                double part[256];
                for (int tid = 0; tid < 256; ++tid) {
                    double p = 0.0;
                    for (int v = tid; v < Vm; v += 256) {
                        const double gm = sG ? m->alpha[mod] + sG[go + k * Vm + v] : gam[v];
                        if (sG) gam[v] = gm;
                        p += gm;
                    }
                    part[tid] = p;
                }
                double sh[4];
                for (int w = 0; w < 4; ++w) sh[w] = tw_wave_sum(part + 64 * w);
                const double cs = sh[0] + sh[1] + sh[2] + sh[3];
                const double pcs = tw_digamma(cs);
                for (int v = 0; v < Vm; ++v) {
                    const double el = tw_digamma(gam[v]) - pcs;
                    El[v] = el;
                    m->expE[go + k * Vm + v] = ar_exp(el);
                    if (m->phi) m->phi[go + (size_t)k * Vm + v] = gam[v] / cs;
                }
            } else {
                const int SJ = tw_SJ(m, mod), nf = m->n_feat[mod], ao = tw_aoff(m, mod);
                const size_t mg = tw_mgoff(m, mod);
                const int32_t* feat = m->features + tw_foff(m, mod);
                double* gam = m->gamma + mg + (size_t)k * SJ; double* El = m->Elnphi + mg + (size_t)k * SJ;
                if (sG) for (int e = 0; e < SJ; ++e) {
                    int jj = e, i = 0;
                    while (jj >= m->J[ao + i]) { jj -= m->J[ao + i]; ++i; }
                    double s = m->alpha[ao + i];
                    for (int v = 0; v < Vm; ++v) if (feat[(size_t)i * Vm + v] == jj) s += sG[go + k * Vm + v];
                    gam[e] = s;
                }
                for (int e = 0; e < SJ; ++e) {
                    int jj = e, i = 0, jo = 0;
                    while (jj >= m->J[ao + i]) { jj -= m->J[ao + i]; jo += m->J[ao + i]; ++i; }
                    double cs = 0.0;
                    for (int j = 0; j < m->J[ao + i]; ++j) cs += gam[jo + j];
                    El[e] = tw_digamma(gam[e]) - tw_digamma(cs);
                }
                for (int v = 0; v < Vm; ++v) {
                    double se = 0.0; int jo = 0;
                    for (int i = 0; i < nf; ++i) { se += El[jo + feat[(size_t)i * Vm + v]]; jo += m->J[ao + i]; }
                    m->expE[go + k * Vm + v] = ar_exp(se);
                }
            }
        }
    }
}

/* ---- fitdoc! for every document, in the order of k_ctm_estep: theta phase (zeta, theta, sumtheta, gamma slabs), then
 * the two LD_MMA solves (MMCTM.jl:450-455 -> :172-198, :156-170, :127-143; IMMCTM.jl:430-435) ------------------------- */
/* sG: [GT] gamma statistics out.  theta (normalised) is stored into m->theta. */
/* The theta phase of the fused pass over ROWS OF COUNTS (k_ctm_theta_dense of ctm.hip; dense corpora): one launch per modality, 16 lanes
 * per document whatever sum K is, four documents per wave step, lane l owns the terms l, 16 + l, ... of every document it meets -- the
 * gamma statistics of its terms are summed per lane over the documents of the wave (in step order), reach the wave's slab one document
 * group at a time, the block's waves in order; sum theta: per lane over its slots in order, then lane k adds the 16 lanes' values in four
 * interleaved chains; zeta: the 16-lane tree with the modality's topics in lanes 0..K_m-1.  Same per-element operations as the slab
 * version below (MMCTM.jl:172-198, 110-117). */
static void tw_theta_dense(orc_ctm* m, double* sumth, double* partial)
{
    const int D = m->D, M = m->M, MK = m->MK, NW = m->waves_e, grid = m->grid_e, GT = tw_GT(m);
    for (int mod = 0; mod < M; ++mod) {
        const int Km = m->K[mod], Vm = m->V[mod], off = tw_koff(m, mod), go = tw_goff(m, mod);
        const int SL = Vm <= 32 ? 2 : (Vm <= 48 ? 3 : (Vm <= 96 ? 6 : 8)), Vp = 16 * SL;
        const int64_t* dp = m->doc_ptr + (size_t)mod * (D + 1);
        const int64_t estart = dp[0];
        const double* tb = m->expE + go;
        double* slabs = (double*)malloc(sizeof(double) * (size_t)NW * Km * Vm);
        double* st = (double*)malloc(sizeof(double) * 64 * (size_t)SL * Km);            /* [lane][slot][k]: the registers of a wave */
        double* cnt = (double*)malloc(sizeof(double) * (size_t)Vp);
        int64_t* pos = (int64_t*)malloc(sizeof(int64_t) * (size_t)Vp);
        for (int b = 0; b < grid; ++b) {
            for (size_t i = 0; i < (size_t)NW * Km * Vm; ++i) slabs[i] = 0.0;
            for (int w = 0; w < NW; ++w) {
                for (size_t i = 0; i < 64 * (size_t)SL * Km; ++i) st[i] = 0.0;
                for (int base = (b * NW + w) * 4; base < D; base += grid * NW * 4) {
                    for (int g = 0; g < 4; ++g) {
                        const int d = base + g;
                        if (d >= D) continue;
                        const double* lam = m->lambda + (size_t)MK * d + off; const double* nu = m->nu + (size_t)MK * d + off;
                        double t[16], a[16], mx = -1e300;
                        for (int l = 0; l < 16; ++l) t[l] = l < Km ? ar_exp(lam[l] + 0.5 * nu[l]) : 0.0;
                        m->zeta[mod + (size_t)M * d] = tw_group_sum(t, 16);
                        for (int kk = 0; kk < Km; ++kk) mx = fmax(mx, lam[kk]);
                        for (int kk = 0; kk < Km; ++kk) a[kk] = ar_exp(lam[kk] - mx);
                        for (int v = 0; v < Vp; ++v) { cnt[v] = 0.0; pos[v] = -1; }
                        for (int64_t e = dp[d]; e < dp[d + 1]; ++e) { cnt[m->term[e]] = (double)m->count[e]; pos[m->term[e]] = e; }
                        double acc[16][16];
                        for (int l = 0; l < 16; ++l) {
                            for (int kk = 0; kk < Km; ++kk) acc[l][kk] = 0.0;
                            for (int q = 0; q < SL; ++q) {
                                const int v = q * 16 + l;
                                double ek[16], s = 0.0;
                                for (int kk = 0; kk < Km; ++kk) { ek[kk] = a[kk] * (v < Vm ? tb[kk * Vm + v] : 1.0); s += ek[kk]; }
                                const double inv = 1.0 / s, r = cnt[v] * inv;
                                double* stq = st + ((size_t)(g * 16 + l) * SL + q) * Km;
                                for (int kk = 0; kk < Km; ++kk) { acc[l][kk] = fma(ek[kk], r, acc[l][kk]); stq[kk] = fma(ek[kk], r, stq[kk]); }
                                if (pos[v] >= 0) {
                                    double* th = m->theta + tw_toff(m, mod) + (size_t)(pos[v] - estart) * Km;
                                    for (int kk = 0; kk < Km; ++kk) th[kk] = ek[kk] * inv;
                                }
                            }
                        }
                        for (int kk = 0; kk < Km; ++kk) {
                            double r4[4] = {0.0, 0.0, 0.0, 0.0};
                            for (int j = 0; j < 16; j += 4) for (int c = 0; c < 4; ++c) r4[c] += acc[j + c][kk];
                            sumth[(size_t)d * MK + off + kk] = (r4[0] + r4[1]) + (r4[2] + r4[3]);
                        }
                    }
                }
                /* the wave's four document groups meet across its rows: (g0 + g2) + (g1 + g3) */
                double* slab = slabs + (size_t)w * Km * Vm;
                for (int l = 0; l < 16; ++l)
                    for (int q = 0; q < SL; ++q) {
                        const int v = q * 16 + l;
                        if (v >= Vm) continue;
                        const double* s0 = st + ((size_t)(0 * 16 + l) * SL + q) * Km, * s1 = st + ((size_t)(1 * 16 + l) * SL + q) * Km;
                        const double* s2 = st + ((size_t)(2 * 16 + l) * SL + q) * Km, * s3 = st + ((size_t)(3 * 16 + l) * SL + q) * Km;
                        for (int kk = 0; kk < Km; ++kk) slab[(size_t)kk * Vm + v] = (s0[kk] + s2[kk]) + (s1[kk] + s3[kk]);
                    }
            }
            for (int i = 0; i < Km * Vm; ++i) {
                double s = 0.0;
                for (int w = 0; w < NW; ++w) s += slabs[(size_t)w * Km * Vm + i];
                partial[(size_t)b * GT + go + i] = s;
            }
        }
        free(slabs); free(st); free(cnt); free(pos);
    }
}

static void tw_estep(orc_ctm* m, double* sG, int dense)
{
    const int D = m->D, M = m->M, MK = m->MK, L = m->L, G = 64 / L, NW = m->waves_e, grid = m->grid_e, GT = tw_GT(m);
    double* sumth = (double*)calloc((size_t)D * MK + 1, sizeof(double));
    double* slabs = (double*)malloc(sizeof(double) * (size_t)NW * GT);
    double* partial = (double*)malloc(sizeof(double) * (size_t)grid * GT);
    double* pn = (double*)malloc(sizeof(double) * 64 * 32);        /* [lane][k] */
    if (dense) tw_theta_dense(m, sumth, partial);
    else
    for (int b = 0; b < grid; ++b) {
        for (size_t i = 0; i < (size_t)NW * GT; ++i) slabs[i] = 0.0;
        for (int w = 0; w < NW; ++w) {
            double* slab = slabs + (size_t)w * GT;
            for (int base = (b * NW + w) * G; base < D; base += grid * NW * G) {
                double av[4][64];                        /* a_k = exp(lambda - max) per document of the wave */
                for (int g = 0; g < G; ++g) {
                    const int d = base + g;
                    if (d >= D) continue;
                    const double* lam = m->lambda + (size_t)MK * d; const double* nu = m->nu + (size_t)MK * d;
                    /* update_ζ!: lanes of the other modalities contribute 0 to the group tree */
                    for (int mod = 0; mod < M; ++mod) {
                        double t[64];
                        const int off = tw_koff(m, mod);
                        for (int l = 0; l < L; ++l) t[l] = (l >= off && l < off + m->K[mod]) ? ar_exp(lam[l] + 0.5 * nu[l]) : 0.0;
                        m->zeta[mod + (size_t)M * d] = tw_group_sum(t, L);
                        double mx = -1e300;
                        for (int kk = 0; kk < m->K[mod]; ++kk) mx = fmax(mx, lam[off + kk]);
                        for (int kk = 0; kk < m->K[mod]; ++kk) av[g][off + kk] = ar_exp(lam[off + kk] - mx);
                    }
                }
                for (int mod = 0; mod < M; ++mod) {
                    const int Km = m->K[mod], Vm = m->V[mod], off = tw_koff(m, mod), go = tw_goff(m, mod);
                    const int64_t* dp = m->doc_ptr + (size_t)mod * (D + 1);
                    const int64_t estart = dp[0];
                    const double* tb = m->expE + go;
                    double acc[64][32];
                    int Wg[4]; int64_t startg[4]; int Wmax = 0;
                    for (int g = 0; g < G; ++g) {
                        const int d = base + g;
                        Wg[g] = d < D ? (int)(dp[d + 1] - dp[d]) : 0; startg[g] = d < D ? dp[d] : 0;
                        if (Wg[g] > Wmax) Wmax = Wg[g];
                    }
                    for (int ln = 0; ln < 64; ++ln) for (int kk = 0; kk < Km; ++kk) acc[ln][kk] = 0.0;
                    for (int w0 = 0; w0 < Wmax; w0 += L) {
                        int act[64], term[64];
                        for (int ln = 0; ln < 64; ++ln) {
                            const int g = ln / L, l = ln % L, wi = w0 + l;
                            act[ln] = g < G && wi < Wg[g];
                            if (!act[ln]) { for (int kk = 0; kk < Km; ++kk) pn[ln * 32 + kk] = 0.0; continue; }
                            const int64_t e = startg[g] + wi;
                            const int v = m->term[e]; const double cnt = (double)m->count[e];
                            term[ln] = v;
                            double ek[32], s = 0.0;
                            for (int kk = 0; kk < Km; ++kk) { ek[kk] = av[g][off + kk] * tb[kk * Vm + v]; s += ek[kk]; }
                            const double inv = 1.0 / s, r = cnt * inv;
                            double* th = m->theta + tw_toff(m, mod) + (size_t)(e - estart) * Km;
                            for (int kk = 0; kk < Km; ++kk) {
                                const double p = ek[kk] * r;
                                pn[ln * 32 + kk] = p;
                                acc[ln][kk] += p;
                                th[kk] = ek[kk] * inv;
                            }
                        }
                        /* one ds_add_f64 per topic: same-address lanes are applied in ascending lane order (measured:
                         * profiles/experiments/r02_lds_atomic_order.txt) */
                        for (int kk = 0; kk < Km; ++kk)
                            for (int ln = 0; ln < 64; ++ln) if (act[ln]) slab[go + kk * Vm + term[ln]] += pn[ln * 32 + kk];
                    }
                    for (int g = 0; g < G; ++g) {
                        const int d = base + g;
                        if (d >= D) continue;
                        for (int kk = 0; kk < Km; ++kk) {
                            double t[64];
                            for (int l = 0; l < L; ++l) t[l] = acc[g * L + l][kk];
                            sumth[(size_t)d * MK + off + kk] = tw_group_sum(t, L);
                        }
                    }
                }
            }
        }
        for (int i = 0; i < GT; ++i) {
            double s = 0.0;
            for (int w = 0; w < NW; ++w) s += slabs[(size_t)w * GT + i];
            partial[(size_t)b * GT + i] = s;
        }
    }
    tw_reduce_partials(partial, grid, GT, sG);
    /* solve phase: nu from the old lambda, then lambda with the new nu */
    for (int d = 0; d < D; ++d) {
        double c[64], x[64];
        for (int mod = 0; mod < M; ++mod) {
            const int64_t* dp = m->doc_ptr + (size_t)mod * (D + 1);
            double Nd = 0.0;
            for (int64_t e = dp[d]; e < dp[d + 1]; ++e) Nd += m->count[e];
            const double cl = Nd / m->zeta[mod + (size_t)M * d];
            for (int kk = 0; kk < m->K[mod]; ++kk) c[tw_koff(m, mod) + kk] = cl;
        }
        double* lam = m->lambda + (size_t)MK * d; double* nu = m->nu + (size_t)MK * d;
        /* Ls: lanes per document in the solve phase; cpl: coordinates per lane (k_ctm_solve_cpl: Ls * cpl = sum K) */
        tw_obj o = { MK, m->cpl > 1 ? MK : (m->Ls > 0 ? m->Ls : L), m->cpl, lam, c, sumth + (size_t)d * MK, m->mu, m->invSigma };
        memcpy(x, nu, sizeof(double) * (size_t)MK);
        int nev = tw_mma(&o, tw_nu_eval, x, 1, m->nu_lower, m->xtol_rel, m->xtol_abs, m->xtol_rule, m->max_eval);
        memcpy(nu, x, sizeof(double) * (size_t)MK);
        if (m->nev_nu) m->nev_nu[d] = nev;
        if (nev < 0) m->n_solver_cap++; else m->n_eval_nu += nev;
        o.other = nu;
        memcpy(x, lam, sizeof(double) * (size_t)MK);
        nev = tw_mma(&o, tw_lam_eval, x, 0, 0.0, m->xtol_rel, m->xtol_abs, m->xtol_rule, m->max_eval);
        memcpy(lam, x, sizeof(double) * (size_t)MK);
        if (m->nev_lambda) m->nev_lambda[d] = nev;
        if (nev < 0) m->n_solver_cap++; else m->n_eval_lambda += nev;
    }
    free(sumth); free(slabs); free(partial); free(pn);
}

void orc_twin_estep(orc_ctm* m, double* sG) { tw_estep(m, sG, 0); }      /* the theta phase by slabs: stage calls, frozen-topic passes */
void orc_twin_estep_fused(orc_ctm* m, double* sG) { tw_estep(m, sG, m->tdense); }      /* ... as the fused pass of the device handle runs it */

/* ---- sum lambda, sum nu, sum lambda lambda' : k_ctm_moments + k_reduce_partials ------------------------------------ */
/* mom: [MK | MK | MK*MK] */
void orc_twin_moments(const orc_ctm* m, double* mom)
{
    const int D = m->D, MK = m->MK, nb = m->grid_m, n = 2 * MK + MK * MK, T = 32;
    double* part = (double*)calloc((size_t)nb * n, sizeof(double));
    const int per = (D + nb - 1) / nb;
    for (int b = 0; b < nb; ++b) {
        const int d0 = b * per, d1 = D < d0 + per ? D : d0 + per;
        for (int e = 0; e < n; ++e) {
            double acc = 0.0;
            for (int t0 = d0; t0 < d1; t0 += T) {
                const int nt = (d1 - t0) < T ? (d1 - t0) : T;
                double s[4] = {0.0, 0.0, 0.0, 0.0};
                for (int dd = 0; dd < T; ++dd) {
                    const int in = dd < nt;
                    const double* lam = m->lambda + (size_t)MK * (t0 + dd); const double* nu = m->nu + (size_t)MK * (t0 + dd);
                    if (e < MK) s[dd & 3] += in ? lam[e] : 0.0;
                    else if (e < 2 * MK) s[dd & 3] += in ? nu[e - MK] : 0.0;
                    else {
                        const int a = (e - 2 * MK) % MK, bq = (e - 2 * MK) / MK;
                        s[dd & 3] = fma(in ? lam[a] : 0.0, in ? lam[bq] : 0.0, s[dd & 3]);
                    }
                }
                acc += (s[0] + s[1]) + (s[2] + s[3]);
            }
            part[(size_t)b * n + e] = acc;
        }
    }
    tw_reduce_partials(part, nb, n, mom);
    free(part);
}

/* ---- update_μ! / update_Σ! from the raw moments: ctm_gauss_mstep + block_inverse_wide (MMCTM.jl:200-212) ------------- */
int orc_twin_gauss(orc_ctm* m, const double* mom, int do_sigma)
{
    const int n = m->MK;
    const double Dg = (double)m->D;
    const double* sLam = mom; const double* sNu = mom + n; const double* sLL = mom + 2 * n;
    for (int i = 0; i < n; ++i) m->mu[i] = sLam[i] / Dg;
    if (!do_sigma) return 0;
    double* A = (double*)malloc(sizeof(double) * (size_t)n * n * 2);
    double* Ai = A + (size_t)n * n;
    for (int e = 0; e < n * n; ++e) {
        const int i = e % n, j = e / n;
        const double mi = m->mu[i], mj = m->mu[j];
        double v = sLL[e] - mi * sLam[j] - mj * sLam[i] + Dg * mi * mj;
        if (i == j) v += sNu[i];
        v /= Dg;
        m->Sigma[e] = v; A[i * n + j] = v;
    }
    for (int i = 0; i < n * n; ++i) Ai[i] = 0.0;
    for (int i = 0; i < n; ++i) Ai[i * n + i] = 1.0;
    int singular = 0;
    double* scol = (double*)malloc(sizeof(double) * (size_t)n);
    for (int c = 0; c < n; ++c) {
        int p = c; double best = fabs(A[c * n + c]);
        for (int r = c + 1; r < n; ++r) { const double a = fabs(A[r * n + c]); if (a > best) { best = a; p = r; } }
        if (best == 0.0) singular = 1;      /* (a NaN column is not an error upstream, nor on the device: block_inverse_wide) */
        const double piv = A[p * n + c];
        for (int q = 0; q < 2; ++q) {
            double* Mx = q ? Ai : A;
            for (int j = 0; j < n; ++j) {
                const double top = Mx[c * n + j], low = Mx[p * n + j];
                Mx[c * n + j] = low / piv;
                if (p != c) Mx[p * n + j] = top;
            }
        }
        for (int r = 0; r < n; ++r) scol[r] = A[r * n + c];
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = scol[r];
            for (int j = 0; j < n; ++j) {
                const double ac = A[c * n + j], ic = Ai[c * n + j];
                A[r * n + j] = (j == c) ? 0.0 : A[r * n + j] - f * ac;
                Ai[r * n + j] -= f * ic;
            }
        }
    }
    for (int e = 0; e < n * n; ++e) { const int i = e % n, j = e / n; m->invSigma[e] = Ai[i * n + j]; }
    free(scol); free(A);
    return singular ? -2 : 0;
}

/* exp table of the theta phase from UPLOADED Elnphi (fit_heldout copies gamma and Elnphi, MMCTM.jl:561-562): k_ctm_tables_from_Elnphi */
void orc_twin_tables_from_Elnphi(orc_ctm* m)
{
    for (int mod = 0; mod < m->M; ++mod) {
        const int Km = m->K[mod], Vm = m->V[mod], go = tw_goff(m, mod);
        for (int k = 0; k < Km; ++k) {
            if (!m->n_feat) {
                for (int v = 0; v < Vm; ++v) m->expE[go + k * Vm + v] = ar_exp(m->Elnphi[go + (size_t)k * Vm + v]);
            } else {
                const int SJ = tw_SJ(m, mod), nf = m->n_feat[mod], ao = tw_aoff(m, mod);
                const size_t mg = tw_mgoff(m, mod);
                const int32_t* feat = m->features + tw_foff(m, mod);
                for (int v = 0; v < Vm; ++v) {
                    double se = 0.0; int jo = 0;
                    for (int i = 0; i < nf; ++i) { se += m->Elnphi[mg + (size_t)k * SJ + jo + feat[(size_t)i * Vm + v]]; jo += m->J[ao + i]; }
                    m->expE[go + k * Vm + v] = ar_exp(se);
                }
            }
        }
    }
}

/* one frozen-topic pass in device order: the document loop of transform (flags & 1: theta from phi instead of exp(Elnphi),
 * MMCTM.jl:496-509, 521-528) or of fit_heldout / predict_modality_eta (MMCTM.jl:565-569), optionally update_mu! / update_Sigma!
 * (flags & 2, :530-533).  The exp table must be current (orc_twin_tables_from_Elnphi). */
int orc_twin_infer_pass(orc_ctm* m, int flags)
{
    const int GT = tw_GT(m), n = m->MK;
    double* sG = (double*)malloc(sizeof(double) * ((size_t)GT + 2 * n + (size_t)n * n));
    double* keep = m->expE;
    if (flags & 1) m->expE = m->phi;                    /* MMCTM: phi has the effective [m][k][v] layout */
    orc_twin_estep(m, sG);
    m->expE = keep;
    int rc = 0;
    if (flags & 2) { orc_twin_moments(m, sG + GT); rc = orc_twin_gauss(m, sG + GT, 1); }
    free(sG);
    return rc;
}

/* one pass of fit! up to and including update_γ!/Elnϕ! (MMCTM.jl:462-471), device order */
int orc_twin_pass(orc_ctm* m, int update_sigma)
{
    const int GT = tw_GT(m), n = m->MK;
    double* sG = (double*)malloc(sizeof(double) * ((size_t)GT + 2 * n + (size_t)n * n));
    double* mom = sG + GT;
    tw_estep(m, sG, m->tdense);          /* the fused pass: over rows of counts when the device handle says so (geometry) */
    orc_twin_moments(m, mom);
    const int rc = orc_twin_gauss(m, mom, update_sigma);
    orc_twin_topics(m, sG);
    free(sG);
    return rc;
}

/* the two objectives at (lambda, nu) in the reference's MAXIMISATION form (common.jl:11-36), evaluated by the functions the
 * solves above use: vals = {lambda_objective, nu_objective}; for the known-answer tests (test/common.jl:79-97) */
void orc_twin_objectives(int n, const double* lambda, const double* nu, const double* Ndivzeta, const double* sumtheta,
                         const double* mu, const double* invSigma, double* vals, double* grad_lambda, double* grad_nu)
{
    const int L = n <= 16 ? 16 : (n <= 32 ? 32 : 64);
    double g[64];
    tw_obj o = { n, L, 1, nu, Ndivzeta, sumtheta, mu, invSigma };
    vals[0] = -tw_lam_eval(&o, lambda, g);
    for (int i = 0; i < n; ++i) grad_lambda[i] = -g[i];
    o.other = lambda;
    vals[1] = -tw_nu_eval(&o, nu, g);
    for (int i = 0; i < n; ++i) grad_nu[i] = -g[i];
}

/* ---- vector entry points for the arithmetic tests ------------------------------------------------------------------- */
void orc_ar_exp_vec(int n, const double* x, double* out) { for (int i = 0; i < n; ++i) out[i] = ar_exp(x[i]); }
void orc_ar_log_vec(int n, const double* x, double* out) { for (int i = 0; i < n; ++i) out[i] = ar_log(x[i]); }
void orc_ar_digamma_vec(int n, const double* x, double* out) { for (int i = 0; i < n; ++i) out[i] = ar_digamma_pos(x[i]); }
void orc_ar_exptab_vec(int n, const double* x, double* out) { for (int i = 0; i < n; ++i) out[i] = ar_exp_tab(x[i], tw_exptab); }
void orc_ar_logtab_vec(int n, const double* x, double* out) { for (int i = 0; i < n; ++i) out[i] = ar_log_tab(x[i], tw_logtab); }
void orc_ar_digammatab_vec(int n, const double* x, double* out) { for (int i = 0; i < n; ++i) out[i] = ar_digamma_pos_tab(x[i], tw_logtab); }
