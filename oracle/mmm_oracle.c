/*
 * mmm_oracle.c -- CPU ORACLE (test infrastructure only; see mmm_oracle.h for the contract).
 *
 * Every function cites the reference file:line (under /root/reference/src) it restates.  The code is a
 * deliberately literal, sequential restatement: same update order, same operand order inside sums
 * where the Julia source fixes one, no fusion, no reordering for speed.
 */
#include "mmm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ============================================================================================== */
/* scalar math                                                                                    */
/* ============================================================================================== */

/* SpecialFunctions.jl `digamma(x::Float64)` (un-vendored dependency, Project.toml:16): reflection for
 * x <= 0, recurrence up to x >= 7, then the 8-term asymptotic series in 1/x^2 with coefficients
 * B_{2k}/(2k).  Call sites in the reference: LDA.jl:79,97; MMCTM.jl:218; IMMCTM.jl:192-193. */
double orc_digamma(double x)
{
    double psi = 0.0;
    if (x <= 0.0) {
        psi -= M_PI / tan(M_PI * x);
        x = 1.0 - x;
    }
    if (x < 7.0) {
        int n = 7 - (int)floor(x);
        for (int v = 1; v < n; ++v) psi -= 1.0 / (x + (double)v);
        psi -= 1.0 / x;
        x += (double)n;
    }
    double t = 1.0 / x;
    psi += log(x) - 0.5 * t;
    t *= t;
    /* Horner, highest coefficient first */
    double p = -0.4432598039215686;
    p = p * t + 0.08333333333333333;
    p = p * t + -0.021092796092796094;
    p = p * t + 0.007575757575757576;
    p = p * t + -0.004166666666666667;
    p = p * t + 0.003968253968253968;
    p = p * t + -0.008333333333333333;
    p = p * t + 0.08333333333333333;
    psi -= t * p;
    return psi;
}

void orc_digamma_vec(int n, const double* x, double* out)
{
    for (int i = 0; i < n; ++i) out[i] = orc_digamma(x[i]);
}

/* logabsgamma(x)[1] / lgamma(x) -- LDA.jl:115,121,143,149; common.jl:4,6 */
double orc_lgamma(double x)
{
    int sign;
    return lgamma_r(x, &sign);
}

/* common.jl:1-9 */
double orc_logmvbeta(int n, const double* vals)
{
    double r = 0.0, s = 0.0;
    for (int i = 0; i < n; ++i) { r += orc_lgamma(vals[i]); s += vals[i]; }
    r -= orc_lgamma(s);
    return r;
}

/* ============================================================================================== */
/* dense helper: LU with partial pivoting (Julia `inv`, `logdet` -> LAPACK getrf/getri)           */
/* ============================================================================================== */
int orc_inv_logdet(int n, const double* A, double* Ainv, double* logabsdet, int* sign)
{
    double* lu = (double*)malloc(sizeof(double) * (size_t)n * n);
    int* piv = (int*)malloc(sizeof(int) * (size_t)n);
    if (!lu || !piv) { free(lu); free(piv); return -1; }
    memcpy(lu, A, sizeof(double) * (size_t)n * n);
    int sg = 1, rc = 0;
    double lad = 0.0;
    for (int c = 0; c < n; ++c) {
        int p = c; double best = fabs(lu[c + n * c]);
        for (int r = c + 1; r < n; ++r) { double a = fabs(lu[r + n * c]); if (a > best) { best = a; p = r; } }
        piv[c] = p;
        if (best == 0.0) { rc = -2; break; }
        if (p != c) {
            sg = -sg;
            for (int j = 0; j < n; ++j) { double t = lu[c + n * j]; lu[c + n * j] = lu[p + n * j]; lu[p + n * j] = t; }
        }
        double d = lu[c + n * c];
        if (d < 0) sg = -sg;
        lad += log(fabs(d));
        for (int r = c + 1; r < n; ++r) lu[r + n * c] /= d;
        for (int j = c + 1; j < n; ++j) {
            double f = lu[c + n * j];
            for (int r = c + 1; r < n; ++r) lu[r + n * j] -= lu[r + n * c] * f;
        }
    }
    if (rc == 0 && Ainv) {
        /* solve A X = I column by column */
        for (int j = 0; j < n; ++j) {
            double* x = Ainv + (size_t)n * j;
            for (int i = 0; i < n; ++i) x[i] = (i == j) ? 1.0 : 0.0;
            for (int c = 0; c < n; ++c) { int p = piv[c]; if (p != c) { double t = x[c]; x[c] = x[p]; x[p] = t; } }
            for (int c = 0; c < n; ++c) for (int r = c + 1; r < n; ++r) x[r] -= lu[r + n * c] * x[c];
            for (int c = n - 1; c >= 0; --c) { x[c] /= lu[c + n * c]; for (int r = 0; r < c; ++r) x[r] -= lu[r + n * c] * x[c]; }
        }
    }
    if (logabsdet) *logabsdet = lad;
    if (sign) *sign = sg;
    free(lu); free(piv);
    return rc;
}

/* ============================================================================================== */
/* NLopt LD_MMA, zero nonlinear constraints                                                        */
/* ============================================================================================== */
/*
 * Restatement of the published CCSA/MMA algorithm (Svanberg 2002) as implemented by NLopt's
 * `mma_minimize` (libnlopt is an un-vendored dependency: Project.toml:15 `NLopt = "0.5.1, ~0.6"`;
 * call sites MMCTM.jl:127-143,156-170; IMMCTM.jl:107-139).  With m = 0 the dual problem is trivial:
 * each coordinate of the separable convex approximation
 *      g(x+dx) = f + sum_j (dfdx_j sigma_j^2 dx_j + (|dfdx_j| sigma_j + rho/2) dx_j^2) / (sigma_j^2 - dx_j^2)
 * is minimised in closed form.  sigma_j = 1 (a bound is infinite), rho = 1 initially; the inner loop
 * raises rho until the approximation is conservative; the outer loop shrinks rho and adapts sigma by the
 * sign of successive steps.  Stop: NLopt's x-tolerance test on (xcur, xprev).
 */
#define ORC_MMA_RHOMIN 1e-5

static int orc_stop_x(int n, const double* x, const double* oldx, double xtol_rel, double xtol_abs, int rule)
{
    if (rule == 0) {
        /* NLopt >= 2.7: ||x - oldx||_1 < xtol_rel * ||x||_1, or every |dx_j| < xtol_abs */
        double dn = 0.0, xn = 0.0;
        for (int j = 0; j < n; ++j) { dn += fabs(x[j] - oldx[j]); xn += fabs(x[j]); }
        if (dn < xtol_rel * xn) return 1;
        for (int j = 0; j < n; ++j) if (fabs(x[j] - oldx[j]) >= xtol_abs) return 0;
        return 1;
    }
    /* NLopt <= 2.6: every coordinate passes relstop(old, new, xtol_rel, xtol_abs) */
    for (int j = 0; j < n; ++j) {
        double vold = oldx[j], vnew = x[j];
        if (isinf(vold)) return 0;
        double ad = fabs(vnew - vold);
        if (!(ad < xtol_abs || ad < xtol_rel * (fabs(vnew) + fabs(vold)) * 0.5 || (xtol_rel > 0 && vnew == vold)))
            return 0;
    }
    return 1;
}

/* optional trace of the solver's inner iterations, for tests/test_mma_independent.py: row = [rho, gval, wval, fcur, sigma[n], xcur[n]] */
static double* g_mma_trace = NULL; static int g_mma_trace_cap = 0; static int g_mma_trace_rows = 0;
void orc_mma_set_trace(double* buf, int cap_rows) { g_mma_trace = buf; g_mma_trace_cap = cap_rows; g_mma_trace_rows = 0; }
int orc_mma_trace_rows(void) { return g_mma_trace_rows; }

int orc_mma_minimize(int n, orc_objective f, void* data, const double* lb, const double* ub,
                     double* x, double* minf, double xtol_rel, double xtol_abs, int xtol_rule,
                     int max_eval, int* n_outer)
{
    double* w = (double*)malloc(sizeof(double) * (size_t)n * 6);
    double *sigma = w, *dfdx = w + n, *dfdx_cur = w + 2 * n, *xcur = w + 3 * n, *xprev = w + 4 * n,
           *xprevprev = w + 5 * n;
    int nev = 0, k = 0, capped = 0;
    double rho = 1.0, fcur, fbest;

    for (int j = 0; j < n; ++j) {
        double l = lb ? lb[j] : -HUGE_VAL, u = ub ? ub[j] : HUGE_VAL;
        sigma[j] = (isinf(l) || isinf(u)) ? 1.0 : 0.5 * (u - l);
    }
    fbest = fcur = f(n, x, dfdx, data); ++nev;
    memcpy(xcur, x, sizeof(double) * n);

    for (;;) { /* outer iterations */
        if (max_eval > 0 && nev >= max_eval) { capped = 1; break; }
        if (++k > 1) memcpy(xprevprev, xprev, sizeof(double) * n);
        memcpy(xprev, xcur, sizeof(double) * n);

        for (;;) { /* inner iterations */
            double gval = fbest, wval = 0.0;
            for (int j = 0; j < n; ++j) {
                double l = lb ? lb[j] : -HUGE_VAL, u_b = ub ? ub[j] : HUGE_VAL;
                if (sigma[j] == 0.0) { xcur[j] = x[j]; continue; }
                double g = dfdx[j];
                double sigma2 = sigma[j] * sigma[j];
                double u = g * sigma2;
                double v = fabs(g) * sigma[j] + 0.5 * rho;
                double q = u / (v * sigma[j]);
                double dx = (u / v) / (-1.0 - sqrt(fabs(1.0 - q * q)));
                double xc = x[j] + dx;
                if (xc > u_b) xc = u_b; else if (xc < l) xc = l;
                if (xc > x[j] + 0.9 * sigma[j]) xc = x[j] + 0.9 * sigma[j];
                else if (xc < x[j] - 0.9 * sigma[j]) xc = x[j] - 0.9 * sigma[j];
                xcur[j] = xc;
                dx = xc - x[j];
                double dx2 = dx * dx;
                double denominv = 1.0 / (sigma2 - dx2);
                gval += (g * (sigma2 * dx) + (fabs(g) * sigma[j] + 0.5 * rho) * dx2) * denominv;
                wval += 0.5 * dx2 * denominv;
            }
            fcur = f(n, xcur, dfdx_cur, data); ++nev;
            if (g_mma_trace && g_mma_trace_rows < g_mma_trace_cap) {
                double* row = g_mma_trace + (size_t)g_mma_trace_rows++ * (4 + 2 * (size_t)n);
                row[0] = rho; row[1] = gval; row[2] = wval; row[3] = fcur;
                memcpy(row + 4, sigma, sizeof(double) * n); memcpy(row + 4 + n, xcur, sizeof(double) * n);
            }
            int inner_done = gval >= fcur;
            if (fcur < fbest) {
                fbest = fcur;
                memcpy(x, xcur, sizeof(double) * n);
                memcpy(dfdx, dfdx_cur, sizeof(double) * n);
            }
            if (max_eval > 0 && nev >= max_eval) { capped = 1; break; }
            if (inner_done) break;
            if (fcur > gval) {
                double a = 10.0 * rho, b = 1.1 * (rho + (fcur - gval) / wval);
                rho = a < b ? a : b;
            }
        }
        if (capped) break;
        if (orc_stop_x(n, xcur, xprev, xtol_rel, xtol_abs, xtol_rule)) break;

        rho = 0.1 * rho > ORC_MMA_RHOMIN ? 0.1 * rho : ORC_MMA_RHOMIN;
        if (k > 1) {
            for (int j = 0; j < n; ++j) {
                double dx2 = (xcur[j] - xprev[j]) * (xprev[j] - xprevprev[j]);
                double gam = dx2 < 0 ? 0.7 : (dx2 > 0 ? 1.2 : 1.0);
                sigma[j] *= gam;
                double l = lb ? lb[j] : -HUGE_VAL, u = ub ? ub[j] : HUGE_VAL;
                if (!isinf(u) && !isinf(l)) {
                    if (sigma[j] > 10.0 * (u - l)) sigma[j] = 10.0 * (u - l);
                    if (sigma[j] < 0.01 * (u - l)) sigma[j] = 0.01 * (u - l);
                }
            }
        }
    }
    if (minf) *minf = fbest;
    if (n_outer) *n_outer = k;
    free(w);
    return capped ? -1 : nev;
}

/* common.jl:11-23 */
double orc_lambda_objective(int n, const double* lambda, double* grad, const double* nu,
                            const double* Ndivzeta, const double* sumtheta, const double* mu,
                            const double* invSigma)
{
    double* diff = (double*)malloc(sizeof(double) * (size_t)n * 3);
    double* Ee = diff + n; double* Sd = diff + 2 * n;
    for (int i = 0; i < n; ++i) { diff[i] = lambda[i] - mu[i]; Ee[i] = exp(lambda[i] + 0.5 * nu[i]); }
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s += invSigma[i + n * j] * diff[j];
        Sd[i] = s;
    }
    if (grad) for (int i = 0; i < n; ++i) grad[i] = -Sd[i] + sumtheta[i] - Ndivzeta[i] * Ee[i];
    double quad = 0.0, lin = 0.0, ex = 0.0;
    for (int i = 0; i < n; ++i) { quad += diff[i] * Sd[i]; lin += lambda[i] * sumtheta[i]; ex += Ndivzeta[i] * Ee[i]; }
    free(diff);
    return -0.5 * quad + lin - ex;
}

/* common.jl:25-36 */
double orc_nu_objective(int n, const double* nu, double* grad, const double* lambda,
                        const double* Ndivzeta, const double* mu, const double* invSigma)
{
    (void)mu;
    double tr = 0.0, ex = 0.0, lg = 0.0;
    for (int i = 0; i < n; ++i) {
        double Ee = exp(lambda[i] + 0.5 * nu[i]);
        if (grad) grad[i] = -0.5 * invSigma[i + n * i] - (Ndivzeta[i] / 2.0) * Ee + (1.0 / (2.0 * nu[i]));
        tr += nu[i] * invSigma[i + n * i];
        ex += Ndivzeta[i] * Ee;
        lg += log(nu[i]);
    }
    return -0.5 * tr - ex + lg / 2.0;
}

/* ============================================================================================== */
/* LDA                                                                                            */
/* ============================================================================================== */

/* LDA.jl:78-80 */
void orc_lda_update_Elntheta(int K, int D, const double* gamma, double* Elntheta)
{
    for (int d = 0; d < D; ++d) {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += gamma[k + (size_t)K * d];
        double ds = orc_digamma(s);
        for (int k = 0; k < K; ++k) Elntheta[k + (size_t)K * d] = orc_digamma(gamma[k + (size_t)K * d]) - ds;
    }
}

/* LDA.jl:82-90 */
void orc_lda_update_gamma(int K, int D, double alpha, const int64_t* doc_ptr, const int32_t* count,
                          const double* phi, double* gamma, double* Elntheta)
{
    for (int d = 0; d < D; ++d) {
        const double* ph = phi + (size_t)K * doc_ptr[d];
        int64_t W = doc_ptr[d + 1] - doc_ptr[d];
        for (int k = 0; k < K; ++k) {
            double s = 0.0;
            for (int64_t w = 0; w < W; ++w) s += ph[k + (size_t)K * w] * (double)count[doc_ptr[d] + w];
            gamma[k + (size_t)K * d] = alpha + s;
        }
    }
    orc_lda_update_Elntheta(K, D, gamma, Elntheta);
}

/* LDA.jl:69-76 */
void orc_lda_update_phi(int K, int D, int V, const int64_t* doc_ptr, const int32_t* term,
                        const double* Elntheta, const double* Elnbeta, double* phi)
{
    for (int d = 0; d < D; ++d) {
        double* ph = phi + (size_t)K * doc_ptr[d];
        int64_t W = doc_ptr[d + 1] - doc_ptr[d];
        for (int64_t w = 0; w < W; ++w) {
            int v = term[doc_ptr[d] + w];
            double s = 0.0;
            for (int k = 0; k < K; ++k) {
                double e = exp(Elntheta[k + (size_t)K * d] + Elnbeta[v + (size_t)V * k]);
                ph[k + (size_t)K * w] = e; s += e;
            }
            for (int k = 0; k < K; ++k) ph[k + (size_t)K * w] /= s;
        }
    }
}

/* LDA.jl:96-98 */
void orc_lda_update_Elnbeta(int V, int K, const double* lambda, double* Elnbeta)
{
    for (int k = 0; k < K; ++k) {
        double s = 0.0;
        for (int v = 0; v < V; ++v) s += lambda[v + (size_t)V * k];
        double ds = orc_digamma(s);
        for (int v = 0; v < V; ++v) Elnbeta[v + (size_t)V * k] = orc_digamma(lambda[v + (size_t)V * k]) - ds;
    }
}

/* LDA.jl:100-108 */
void orc_lda_update_lambda(int K, int D, int V, double eta, const int64_t* doc_ptr,
                           const int32_t* term, const int32_t* count, const double* phi,
                           double* lambda, double* Elnbeta)
{
    for (size_t i = 0; i < (size_t)V * K; ++i) lambda[i] = eta;
    for (int d = 0; d < D; ++d) {
        const double* ph = phi + (size_t)K * doc_ptr[d];
        int64_t W = doc_ptr[d + 1] - doc_ptr[d];
        for (int64_t w = 0; w < W; ++w) {
            int v = term[doc_ptr[d] + w];
            double n = (double)count[doc_ptr[d] + w];
            for (int k = 0; k < K; ++k) lambda[v + (size_t)V * k] += ph[k + (size_t)K * w] * n;
        }
    }
    orc_lda_update_Elnbeta(V, K, lambda, Elnbeta);
}

/* LDA.jl:110-112 */
void orc_lda_update_beta(int V, int K, const double* lambda, double* beta)
{
    for (int k = 0; k < K; ++k) {
        double s = 0.0;
        for (int v = 0; v < V; ++v) s += lambda[v + (size_t)V * k];
        for (int v = 0; v < V; ++v) beta[v + (size_t)V * k] = lambda[v + (size_t)V * k] / s;
    }
}

/* LDA.jl:92-94 */
void orc_lda_update_theta(int K, int D, const double* gamma, double* theta)
{
    for (int d = 0; d < D; ++d) {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += gamma[k + (size_t)K * d];
        for (int k = 0; k < K; ++k) theta[k + (size_t)K * d] = gamma[k + (size_t)K * d] / s;
    }
}

/* LDA.jl:174-188 */
double orc_lda_loglik(int K, int D, int V, const int64_t* doc_ptr, const int32_t* term,
                      const int32_t* count, const double* theta, const double* beta)
{
    double ll = 0.0; int64_t N = 0;
    for (int d = 0; d < D; ++d) {
        for (int64_t e = doc_ptr[d]; e < doc_ptr[d + 1]; ++e) {
            int v = term[e];
            N += count[e];
            double p = 0.0;
            for (int k = 0; k < K; ++k) p += theta[k + (size_t)K * d] * beta[v + (size_t)V * k];
            ll += (double)count[e] * log(p);
        }
    }
    return ll / (double)N;
}

/* LDA.jl:114-172 */
double orc_lda_elbo(int K, int D, int V, double alpha, double eta, const int64_t* doc_ptr,
                    const int32_t* term, const int32_t* count, const double* lambda,
                    const double* Elnbeta, const double* gamma, const double* Elntheta,
                    const double* phi, double* terms)
{
    double t[7];
    /* ElnPbeta :114-118 */
    double s = 0.0;
    for (size_t i = 0; i < (size_t)V * K; ++i) s += Elnbeta[i];
    t[0] = K * (orc_lgamma(V * eta) - V * orc_lgamma(eta)) + (eta - 1) * s;
    /* ElnPtheta :120-124 */
    s = 0.0;
    for (size_t i = 0; i < (size_t)K * D; ++i) s += Elntheta[i];
    t[1] = D * (orc_lgamma(K * alpha) - K * orc_lgamma(alpha)) + (alpha - 1) * s;
    /* ElnPZ :126-132, ElnPX :134-140, ElnQZ :154-160 */
    double pz = 0.0, px = 0.0, qz = 0.0;
    for (int d = 0; d < D; ++d) {
        const double* ph = phi + (size_t)K * doc_ptr[d];
        int64_t W = doc_ptr[d + 1] - doc_ptr[d];
        for (int64_t w = 0; w < W; ++w) {
            int v = term[doc_ptr[d] + w];
            double n = (double)count[doc_ptr[d] + w];
            for (int k = 0; k < K; ++k) {
                double p = ph[k + (size_t)K * w];
                pz += p * Elntheta[k + (size_t)K * d] * n;
                px += p * Elnbeta[v + (size_t)V * k] * n;
                qz += log(pow(p, p));
            }
        }
    }
    t[2] = pz; t[3] = px; t[6] = qz;
    /* ElnQbeta :142-146 */
    double a = 0.0, b = 0.0, c = 0.0;
    for (int k = 0; k < K; ++k) {
        double cs = 0.0;
        for (int v = 0; v < V; ++v) {
            double l = lambda[v + (size_t)V * k];
            a += orc_lgamma(l); cs += l; c += (l - 1) * Elnbeta[v + (size_t)V * k];
        }
        b += orc_lgamma(cs);
    }
    t[4] = a - b - c;
    /* ElnQtheta :148-152 */
    a = b = c = 0.0;
    for (int d = 0; d < D; ++d) {
        double cs = 0.0;
        for (int k = 0; k < K; ++k) {
            double g = gamma[k + (size_t)K * d];
            a += orc_lgamma(g); cs += g; c += (g - 1) * Elntheta[k + (size_t)K * d];
        }
        b += orc_lgamma(cs);
    }
    t[5] = a - b - c;
    if (terms) memcpy(terms, t, sizeof t);
    return t[0] + t[1] + t[2] + t[3] - t[4] - t[5] - t[6];
}

/* LDA.jl:24-54 (constructor state) + LDA.jl:198-224 (fit!) */
int orc_lda_fit(int D, int V, int K, double alpha, double eta, const int64_t* doc_ptr,
                const int32_t* term, const int32_t* count, int maxiter, double tol,
                double* lambda, double* gamma, double* Elntheta, double* theta, double* Elnbeta,
                double* beta, double* phi, double* ll_hist, int* n_iter, int* converged,
                double* elbo)
{
    int64_t nnz = doc_ptr[D];
    orc_lda_update_Elnbeta(V, K, lambda, Elnbeta);                 /* :36-39 */
    for (size_t i = 0; i < (size_t)K * D; ++i) gamma[i] = 1.0;     /* :41 */
    orc_lda_update_Elntheta(K, D, gamma, Elntheta);                /* :44 */
    for (size_t i = 0; i < (size_t)K * nnz; ++i) phi[i] = 1.0 / K; /* :46-49 */
    *converged = 0;
    int it = 0;
    for (int iter = 1; iter <= maxiter; ++iter) {
        orc_lda_update_gamma(K, D, alpha, doc_ptr, count, phi, gamma, Elntheta);
        orc_lda_update_phi(K, D, V, doc_ptr, term, Elntheta, Elnbeta, phi);
        orc_lda_update_lambda(K, D, V, eta, doc_ptr, term, count, phi, lambda, Elnbeta);
        orc_lda_update_beta(V, K, lambda, beta);
        orc_lda_update_theta(K, D, gamma, theta);
        ll_hist[it++] = orc_lda_loglik(K, D, V, doc_ptr, term, count, theta, beta);
        if (it > 10) { /* common.jl:53-56 */
            double rel = fabs(ll_hist[it - 2] - ll_hist[it - 1]) / fabs(ll_hist[it - 1]);
            if (rel < tol) { *converged = 1; break; }
        }
    }
    *n_iter = it;
    if (elbo) *elbo = orc_lda_elbo(K, D, V, alpha, eta, doc_ptr, term, count, lambda, Elnbeta, gamma, Elntheta, phi, NULL);
    return 0;
}

/* unsmoothed_update_ϕ!  LDA.jl:226-231: phi[d] = exp.(Elntheta[:, d]) .* beta[X[d][:, 1], :]' normalised per term */
void orc_lda_unsmoothed_update_phi(int K, int D, int V, const int64_t* doc_ptr, const int32_t* term,
                                   const double* Elntheta, const double* beta, double* phi)
{
    for (int d = 0; d < D; ++d) {
        double* ph = phi + (size_t)K * doc_ptr[d];
        int64_t W = doc_ptr[d + 1] - doc_ptr[d];
        for (int64_t w = 0; w < W; ++w) {
            int v = term[doc_ptr[d] + w];
            double s = 0.0;
            for (int k = 0; k < K; ++k) {
                double e = exp(Elntheta[k + (size_t)K * d]) * beta[v + (size_t)V * k];
                ph[k + (size_t)K * w] = e; s += e;
            }
            for (int k = 0; k < K; ++k) ph[k + (size_t)K * w] /= s;
        }
    }
}

/* The loops of transform (LDA.jl:233-263, unsmoothed = 1, uses beta) and fit_heldout (LDA.jl:265-295, unsmoothed = 0,
 * uses Elnbeta) on a freshly constructed model (LDA.jl:41-49: gamma = 1, phi = 1/K) whose topics were copied in. */
int orc_lda_infer(int D, int V, int K, double alpha, const int64_t* doc_ptr, const int32_t* term,
                  const int32_t* count, int unsmoothed, int maxiter, double tol, const double* Elnbeta,
                  const double* beta, double* gamma, double* Elntheta, double* theta, double* phi,
                  double* ll_hist, int* n_iter, int* converged)
{
    int64_t nnz = doc_ptr[D];
    for (size_t i = 0; i < (size_t)K * D; ++i) gamma[i] = 1.0;
    orc_lda_update_Elntheta(K, D, gamma, Elntheta);
    for (size_t i = 0; i < (size_t)K * nnz; ++i) phi[i] = 1.0 / K;
    *converged = 0;
    int it = 0;
    for (int iter = 1; iter <= maxiter; ++iter) {
        orc_lda_update_gamma(K, D, alpha, doc_ptr, count, phi, gamma, Elntheta);
        if (unsmoothed) orc_lda_unsmoothed_update_phi(K, D, V, doc_ptr, term, Elntheta, beta, phi);
        else orc_lda_update_phi(K, D, V, doc_ptr, term, Elntheta, Elnbeta, phi);
        orc_lda_update_theta(K, D, gamma, theta);
        ll_hist[it++] = orc_lda_loglik(K, D, V, doc_ptr, term, count, theta, beta);
        if (it > 10) {
            double rel = fabs(ll_hist[it - 2] - ll_hist[it - 1]) / fabs(ll_hist[it - 1]);
            if (rel < tol) { *converged = 1; break; }
        }
    }
    *n_iter = it;
    return 0;
}

/* ============================================================================================== */
/* ILDA (src/ILDA.jl): lambda[i] is J_i x K column-major at K * sum_{q<i} J_q; features [i*V + v] 0-based */
/* ============================================================================================== */
static size_t ilda_off(int K, const int* J, int i) { size_t o = 0; for (int q = 0; q < i; ++q) o += (size_t)J[q] * K; return o; }

/* ILDA.jl:97-104 */
void orc_ilda_update_Elnbeta(int K, int I, const int* J, const double* lam, double* Eln)
{
    for (int i = 0; i < I; ++i) {
        const double* l = lam + ilda_off(K, J, i); double* e = Eln + ilda_off(K, J, i);
        for (int k = 0; k < K; ++k) {
            double s = 0.0;
            for (int j = 0; j < J[i]; ++j) s += l[j + (size_t)J[i] * k];
            double ps = orc_digamma(s);
            for (int j = 0; j < J[i]; ++j) e[j + (size_t)J[i] * k] = orc_digamma(l[j + (size_t)J[i] * k]) - ps;
        }
    }
}

/* ILDA.jl:128-130 */
void orc_ilda_update_beta(int K, int I, const int* J, const double* lam, double* beta)
{
    for (int i = 0; i < I; ++i) {
        const double* l = lam + ilda_off(K, J, i); double* b = beta + ilda_off(K, J, i);
        for (int k = 0; k < K; ++k) {
            double s = 0.0;
            for (int j = 0; j < J[i]; ++j) s += l[j + (size_t)J[i] * k];
            for (int j = 0; j < J[i]; ++j) b[j + (size_t)J[i] * k] = l[j + (size_t)J[i] * k] / s;
        }
    }
}

/* ILDA.jl:65-79 (smoothed; no max-subtraction, as the reference) and :274-287 (unsmoothed = 1: exp(Elntheta) .* prod beta) */
void orc_ilda_update_phi(int K, int D, int V, int I, const int* J, const int32_t* features, const int64_t* doc_ptr,
                         const int32_t* term, const double* Elntheta, const double* Eln_or_beta, int unsmoothed, double* phi)
{
    for (int d = 0; d < D; ++d) {
        double* ph = phi + (size_t)K * doc_ptr[d];
        int64_t W = doc_ptr[d + 1] - doc_ptr[d];
        for (int64_t w = 0; w < W; ++w) {
            int v = term[doc_ptr[d] + w];
            double s = 0.0;
            for (int k = 0; k < K; ++k) {
                double a = unsmoothed ? exp(Elntheta[k + (size_t)K * d]) : Elntheta[k + (size_t)K * d];
                for (int i = 0; i < I; ++i) {
                    int j = features[(size_t)i * V + v];
                    double t = Eln_or_beta[ilda_off(K, J, i) + j + (size_t)J[i] * k];
                    if (unsmoothed) a *= t; else a += t;
                }
                double e = unsmoothed ? a : exp(a);
                ph[k + (size_t)K * w] = e; s += e;
            }
            for (int k = 0; k < K; ++k) ph[k + (size_t)K * w] /= s;
        }
    }
}

/* ILDA.jl:107-126 */
void orc_ilda_update_lambda(int K, int D, int V, int I, const int* J, const double* eta, const int32_t* features,
                            const int64_t* doc_ptr, const int32_t* term, const int32_t* count, const double* phi,
                            double* lam, double* Eln)
{
    for (int i = 0; i < I; ++i) { double* l = lam + ilda_off(K, J, i); for (size_t q = 0; q < (size_t)J[i] * K; ++q) l[q] = eta[i]; }
    for (int d = 0; d < D; ++d) {
        const double* ph = phi + (size_t)K * doc_ptr[d];
        int64_t W = doc_ptr[d + 1] - doc_ptr[d];
        for (int64_t w = 0; w < W; ++w) {
            int v = term[doc_ptr[d] + w]; double n = (double)count[doc_ptr[d] + w];
            for (int i = 0; i < I; ++i) {
                int j = features[(size_t)i * V + v]; double* l = lam + ilda_off(K, J, i);
                for (int k = 0; k < K; ++k) l[j + (size_t)J[i] * k] += ph[k + (size_t)K * w] * n;
            }
        }
    }
    orc_ilda_update_Elnbeta(K, I, J, lam, Eln);
}

/* ILDA.jl:203-231 */
double orc_ilda_loglik(int K, int D, int V, int I, const int* J, const int32_t* features, const int64_t* doc_ptr,
                       const int32_t* term, const int32_t* count, const double* theta, const double* beta)
{
    double ll = 0.0; int64_t N = 0;
    for (int d = 0; d < D; ++d)
        for (int64_t e = doc_ptr[d]; e < doc_ptr[d + 1]; ++e) {
            int v = term[e]; double pw = 0.0;
            N += count[e];
            for (int k = 0; k < K; ++k) {
                double t = theta[k + (size_t)K * d];
                for (int i = 0; i < I; ++i) t *= beta[ilda_off(K, J, i) + features[(size_t)i * V + v] + (size_t)J[i] * k];
                pw += t;
            }
            ll += (double)count[e] * log(pw);
        }
    return ll / (double)N;
}

/* ILDA.jl:132-201; terms as orc_lda_elbo.  ElnQβ follows the reference literally: `lnq = ...` inside the feature loop
 * (ILDA.jl:175-182) discards every feature but the last. */
double orc_ilda_elbo(int K, int D, int V, int I, const int* J, double alpha, const double* eta, const int32_t* features,
                     const int64_t* doc_ptr, const int32_t* term, const int32_t* count, const double* lam, const double* Eln,
                     const double* gamma, const double* Elntheta, const double* phi, double* terms)
{
    double t[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < I; ++i) {
        const double* e = Eln + ilda_off(K, J, i); double se = 0.0;
        for (size_t q = 0; q < (size_t)J[i] * K; ++q) se += e[q];
        t[0] += K * (orc_lgamma(J[i] * eta[i]) - J[i] * orc_lgamma(eta[i])) + (eta[i] - 1) * se;
    }
    { double s = 0.0; for (size_t q = 0; q < (size_t)K * D; ++q) s += Elntheta[q];
      t[1] = D * (orc_lgamma(K * alpha) - K * orc_lgamma(alpha)) + (alpha - 1) * s; }
    for (int d = 0; d < D; ++d) {
        const double* ph = phi + (size_t)K * doc_ptr[d];
        int64_t W = doc_ptr[d + 1] - doc_ptr[d];
        for (int64_t w = 0; w < W; ++w) {
            int v = term[doc_ptr[d] + w]; double n = (double)count[doc_ptr[d] + w];
            for (int k = 0; k < K; ++k) {
                double p = ph[k + (size_t)K * w];
                t[2] += p * Elntheta[k + (size_t)K * d] * n;
                for (int i = 0; i < I; ++i) t[3] += p * n * Eln[ilda_off(K, J, i) + features[(size_t)i * V + v] + (size_t)J[i] * k];
                t[6] += (p > 0.0) ? p * log(p) : 0.0;                 /* log(ϕ^ϕ), 0^0 = 1 */
            }
        }
    }
    for (int i = 0; i < I; ++i) {
        const double* l = lam + ilda_off(K, J, i); const double* e = Eln + ilda_off(K, J, i);
        double q = 0.0;
        for (int k = 0; k < K; ++k) {
            double cs = 0.0;
            for (int j = 0; j < J[i]; ++j) { double x = l[j + (size_t)J[i] * k]; q += orc_lgamma(x) - (x - 1) * e[j + (size_t)J[i] * k]; cs += x; }
            q -= orc_lgamma(cs);
        }
        t[4] = q;      /* sic: overwritten, not accumulated */
    }
    for (int d = 0; d < D; ++d) {
        double cs = 0.0;
        for (int k = 0; k < K; ++k) { double x = gamma[k + (size_t)K * d]; t[5] += orc_lgamma(x) - (x - 1) * Elntheta[k + (size_t)K * d]; cs += x; }
        t[5] -= orc_lgamma(cs);
    }
    if (terms) memcpy(terms, t, sizeof t);
    return t[0] + t[1] + t[2] + t[3] - t[4] - t[5] - t[6];
}

/* constructor state (ILDA.jl:36-51) + fit! (ILDA.jl:246-272); frozen = 1: the loop of fit_heldout (ILDA.jl:330-348) with the
 * given lambda / Elnbeta / beta held fixed.  lam is in/out (lambda0 on entry). */
int orc_ilda_fit(int D, int V, int K, int I, const int* J, double alpha, const double* eta, const int32_t* features,
                 const int64_t* doc_ptr, const int32_t* term, const int32_t* count, int frozen, int maxiter, double tol,
                 double* lam, double* Eln, double* beta, double* gamma, double* Elntheta, double* theta, double* phi,
                 double* ll_hist, int* n_iter, int* converged, double* elbo)
{
    int64_t nnz = doc_ptr[D];
    if (!frozen) orc_ilda_update_Elnbeta(K, I, J, lam, Eln);
    for (size_t q = 0; q < (size_t)K * D; ++q) gamma[q] = 1.0;
    orc_lda_update_Elntheta(K, D, gamma, Elntheta);
    for (size_t q = 0; q < (size_t)K * nnz; ++q) phi[q] = 1.0 / K;
    *converged = 0; int it = 0;
    for (int iter = 1; iter <= maxiter; ++iter) {
        orc_lda_update_gamma(K, D, alpha, doc_ptr, count, phi, gamma, Elntheta);              /* ILDA.jl:85-93 == LDA's */
        orc_ilda_update_phi(K, D, V, I, J, features, doc_ptr, term, Elntheta, Eln, 0, phi);
        if (!frozen) { orc_ilda_update_lambda(K, D, V, I, J, eta, features, doc_ptr, term, count, phi, lam, Eln); orc_ilda_update_beta(K, I, J, lam, beta); }
        orc_lda_update_theta(K, D, gamma, theta);
        ll_hist[it++] = orc_ilda_loglik(K, D, V, I, J, features, doc_ptr, term, count, theta, beta);
        if (it > 10) {
            double rel = fabs(ll_hist[it - 2] - ll_hist[it - 1]) / fabs(ll_hist[it - 1]);
            if (rel < tol) { *converged = 1; break; }
        }
    }
    *n_iter = it;
    if (elbo) *elbo = orc_ilda_elbo(K, D, V, I, J, alpha, eta, features, doc_ptr, term, count, lam, Eln, gamma, Elntheta, phi, NULL);
    return 0;
}

/* ============================================================================================== */
/* MMCTM / IMMCTM                                                                                 */
/* ============================================================================================== */

static inline int ctm_koff(const orc_ctm* m, int mod) { int o = 0; for (int i = 0; i < mod; ++i) o += m->K[i]; return o; }
static inline int ctm_SJ(const orc_ctm* m, int mod) {
    int o = 0, s = 0; for (int i = 0; i < mod; ++i) o += m->n_feat[i];
    for (int i = 0; i < m->n_feat[mod]; ++i) s += m->J[o + i];
    return s;
}
static inline int ctm_aoff(const orc_ctm* m, int mod) { int o = 0; for (int i = 0; i < mod; ++i) o += m->n_feat[i]; return o; }
static inline size_t ctm_goff(const orc_ctm* m, int mod) {
    size_t o = 0;
    for (int i = 0; i < mod; ++i) o += (size_t)m->K[i] * (m->n_feat ? ctm_SJ(m, i) : m->V[i]);
    return o;
}
static inline size_t ctm_foff(const orc_ctm* m, int mod) { size_t o = 0; for (int i = 0; i < mod; ++i) o += (size_t)m->n_feat[i] * m->V[i]; return o; }
static inline int64_t ctm_estart(const orc_ctm* m, int mod) { return m->doc_ptr[(size_t)mod * (m->D + 1)]; }
static inline size_t ctm_toff(const orc_ctm* m, int mod) {
    size_t o = 0;
    for (int i = 0; i < mod; ++i) o += (size_t)(m->doc_ptr[(size_t)i * (m->D + 1) + m->D] - ctm_estart(m, i)) * m->K[i];
    return o;
}
static inline double* ctm_theta(const orc_ctm* m, int mod, int64_t e) { return m->theta + ctm_toff(m, mod) + (size_t)(e - ctm_estart(m, mod)) * m->K[mod]; }
static inline int64_t ctm_N(const orc_ctm* m, int mod, int d) {
    int64_t n = 0; const int64_t* dp = m->doc_ptr + (size_t)mod * (m->D + 1);
    for (int64_t e = dp[d]; e < dp[d + 1]; ++e) n += m->count[e];
    return n;
}
/* IMMCTM: offset of feature i's block inside a topic's gamma row */
static inline int ctm_joff(const orc_ctm* m, int mod, int i) { int a = ctm_aoff(m, mod), s = 0; for (int q = 0; q < i; ++q) s += m->J[a + q]; return s; }

/* MMCTM.jl:172-181 / IMMCTM.jl:141-150 */
void orc_ctm_update_zeta(orc_ctm* m, int d)
{
    const double* lam = m->lambda + (size_t)m->MK * d; const double* nu = m->nu + (size_t)m->MK * d;
    int start = 0;
    for (int mod = 0; mod < m->M; ++mod) {
        double s = 0.0;
        for (int k = 0; k < m->K[mod]; ++k) s += exp(lam[start + k] + 0.5 * nu[start + k]);
        m->zeta[mod + (size_t)m->M * d] = s;
        start += m->K[mod];
    }
}

/* MMCTM.jl:183-198 / IMMCTM.jl:152-172 */
void orc_ctm_update_theta(orc_ctm* m, int d)
{
    const double* lam = m->lambda + (size_t)m->MK * d;
    int offset = 0;
    for (int mod = 0; mod < m->M; ++mod) {
        const int64_t* dp = m->doc_ptr + (size_t)mod * (m->D + 1);
        int Km = m->K[mod];
        size_t go = ctm_goff(m, mod);
        for (int64_t e = dp[d]; e < dp[d + 1]; ++e) {
            int v = m->term[e];
            double* th = ctm_theta(m, mod, e);
            double s = 0.0;
            for (int k = 0; k < Km; ++k) {
                double t;
                if (!m->n_feat) {
                    t = exp(lam[offset + k] + m->Elnphi[go + (size_t)k * m->V[mod] + v]);
                } else {
                    int SJ = ctm_SJ(m, mod); size_t fo = ctm_foff(m, mod);
                    t = exp(lam[offset + k]);
                    for (int i = 0; i < m->n_feat[mod]; ++i) {
                        int f = m->features[fo + (size_t)i * m->V[mod] + v];
                        t *= exp(m->Elnphi[go + (size_t)k * SJ + ctm_joff(m, mod, i) + f]);
                    }
                }
                th[k] = t; s += t;
            }
            for (int k = 0; k < Km; ++k) th[k] /= s;
        }
        offset += Km;
    }
}

/* MMCTM.jl:110-117 */
void orc_ctm_calc_sumtheta(const orc_ctm* m, int d, double* out)
{
    int offset = 0;
    for (int mod = 0; mod < m->M; ++mod) {
        const int64_t* dp = m->doc_ptr + (size_t)mod * (m->D + 1);
        for (int k = 0; k < m->K[mod]; ++k) {
            double s = 0.0;
            for (int64_t e = dp[d]; e < dp[d + 1]; ++e) s += ctm_theta(m, mod, e)[k] * (double)m->count[e];
            out[offset + k] = s;
        }
        offset += m->K[mod];
    }
}

/* MMCTM.jl:119-125 */
void orc_ctm_calc_Ndivzeta(const orc_ctm* m, int d, double* out)
{
    int offset = 0;
    for (int mod = 0; mod < m->M; ++mod) {
        double c = (double)ctm_N(m, mod, d) / m->zeta[mod + (size_t)m->M * d];
        for (int k = 0; k < m->K[mod]; ++k) out[offset + k] = c;
        offset += m->K[mod];
    }
}

typedef struct { const double *other, *Ndivzeta, *sumtheta, *mu, *invSigma; } ctm_cb;

/* NLopt maximises by minimising the negated objective with a negated gradient */
static double cb_lambda(int n, const double* x, double* grad, void* p)
{
    ctm_cb* c = (ctm_cb*)p;
    double v = orc_lambda_objective(n, x, grad, c->other, c->Ndivzeta, c->sumtheta, c->mu, c->invSigma);
    if (grad) for (int i = 0; i < n; ++i) grad[i] = -grad[i];
    return -v;
}
static double cb_nu(int n, const double* x, double* grad, void* p)
{
    ctm_cb* c = (ctm_cb*)p;
    double v = orc_nu_objective(n, x, grad, c->other, c->Ndivzeta, c->mu, c->invSigma);
    if (grad) for (int i = 0; i < n; ++i) grad[i] = -grad[i];
    return -v;
}

/* MMCTM.jl:156-170 / IMMCTM.jl:125-139 */
void orc_ctm_update_nu(orc_ctm* m, int d)
{
    int n = m->MK;
    double* buf = (double*)malloc(sizeof(double) * (size_t)n * 2);
    double* Ndz = buf; double* lb = buf + n;
    orc_ctm_calc_Ndivzeta(m, d, Ndz);
    for (int i = 0; i < n; ++i) lb[i] = m->nu_lower;
    ctm_cb cb = { m->lambda + (size_t)n * d, Ndz, NULL, m->mu, m->invSigma };
    double minf; int no;
    int nev = orc_mma_minimize(n, cb_nu, &cb, lb, NULL, m->nu + (size_t)n * d, &minf, m->xtol_rel, m->xtol_abs, m->xtol_rule, m->max_eval, &no);
    if (m->nev_nu) m->nev_nu[d] = nev;
    if (nev < 0) m->n_solver_cap++; else m->n_eval_nu += nev;
    free(buf);
}

/* MMCTM.jl:127-143 / IMMCTM.jl:107-123 */
void orc_ctm_update_lambda(orc_ctm* m, int d)
{
    int n = m->MK;
    double* buf = (double*)malloc(sizeof(double) * (size_t)n * 2);
    double* Ndz = buf; double* st = buf + n;
    orc_ctm_calc_Ndivzeta(m, d, Ndz);
    orc_ctm_calc_sumtheta(m, d, st);
    ctm_cb cb = { m->nu + (size_t)n * d, Ndz, st, m->mu, m->invSigma };
    double minf; int no;
    int nev = orc_mma_minimize(n, cb_lambda, &cb, NULL, NULL, m->lambda + (size_t)n * d, &minf, m->xtol_rel, m->xtol_abs, m->xtol_rule, m->max_eval, &no);
    if (m->nev_lambda) m->nev_lambda[d] = nev;
    if (nev < 0) m->n_solver_cap++; else m->n_eval_lambda += nev;
    free(buf);
}

/* MMCTM.jl:450-455 */
void orc_ctm_fitdoc(orc_ctm* m, int d)
{
    orc_ctm_update_zeta(m, d);
    orc_ctm_update_theta(m, d);
    orc_ctm_update_nu(m, d);
    orc_ctm_update_lambda(m, d);
}

void orc_ctm_estep_range(orc_ctm* m, int d0, int d1)
{
    for (int d = d0; d < d1; ++d) orc_ctm_fitdoc(m, d);
}

/* MMCTM.jl:200-202 */
void orc_ctm_update_mu(orc_ctm* m)
{
    int n = m->MK;
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int d = 0; d < m->D; ++d) s += m->lambda[i + (size_t)n * d];
        m->mu[i] = s / m->D;
    }
}

/* MMCTM.jl:204-212 */
int orc_ctm_update_Sigma(orc_ctm* m)
{
    int n = m->MK;
    for (size_t i = 0; i < (size_t)n * n; ++i) m->Sigma[i] = 0.0;
    for (int d = 0; d < m->D; ++d) for (int i = 0; i < n; ++i) m->Sigma[i + (size_t)n * i] += m->nu[i + (size_t)n * d];
    double* diff = (double*)malloc(sizeof(double) * n);
    for (int d = 0; d < m->D; ++d) {
        for (int i = 0; i < n; ++i) diff[i] = m->lambda[i + (size_t)n * d] - m->mu[i];
        for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) m->Sigma[i + (size_t)n * j] += diff[i] * diff[j];
    }
    free(diff);
    for (size_t i = 0; i < (size_t)n * n; ++i) m->Sigma[i] /= m->D;
    return orc_inv_logdet(n, m->Sigma, m->invSigma, NULL, NULL);
}

/* MMCTM.jl:214-222 / IMMCTM.jl:188-197 */
void orc_ctm_update_Elnphi(orc_ctm* m)
{
    for (int mod = 0; mod < m->M; ++mod) {
        size_t go = ctm_goff(m, mod);
        if (!m->n_feat) {
            int V = m->V[mod];
            for (int k = 0; k < m->K[mod]; ++k) {
                const double* g = m->gamma + go + (size_t)k * V; double* e = m->Elnphi + go + (size_t)k * V;
                double s = 0.0; for (int v = 0; v < V; ++v) s += g[v];
                double ds = orc_digamma(s);
                for (int v = 0; v < V; ++v) e[v] = orc_digamma(g[v]) - ds;
            }
        } else {
            int SJ = ctm_SJ(m, mod), ao = ctm_aoff(m, mod);
            for (int k = 0; k < m->K[mod]; ++k) {
                int jo = 0;
                for (int i = 0; i < m->n_feat[mod]; ++i) {
                    int Ji = m->J[ao + i];
                    const double* g = m->gamma + go + (size_t)k * SJ + jo; double* e = m->Elnphi + go + (size_t)k * SJ + jo;
                    double s = 0.0; for (int j = 0; j < Ji; ++j) s += g[j];
                    double ds = orc_digamma(s);
                    for (int j = 0; j < Ji; ++j) e[j] = orc_digamma(g[j]) - ds;
                    jo += Ji;
                }
            }
        }
    }
}

/* MMCTM.jl:224-242 / IMMCTM.jl:199-223 */
void orc_ctm_update_gamma(orc_ctm* m)
{
    for (int mod = 0; mod < m->M; ++mod) {
        size_t go = ctm_goff(m, mod);
        if (!m->n_feat) {
            for (size_t i = 0; i < (size_t)m->K[mod] * m->V[mod]; ++i) m->gamma[go + i] = m->alpha[mod];
        } else {
            int SJ = ctm_SJ(m, mod), ao = ctm_aoff(m, mod);
            for (int k = 0; k < m->K[mod]; ++k) {
                int jo = 0;
                for (int i = 0; i < m->n_feat[mod]; ++i) { for (int j = 0; j < m->J[ao + i]; ++j) m->gamma[go + (size_t)k * SJ + jo + j] = m->alpha[ao + i]; jo += m->J[ao + i]; }
            }
        }
    }
    for (int d = 0; d < m->D; ++d) {
        for (int mod = 0; mod < m->M; ++mod) {
            const int64_t* dp = m->doc_ptr + (size_t)mod * (m->D + 1);
            size_t go = ctm_goff(m, mod);
            for (int64_t e = dp[d]; e < dp[d + 1]; ++e) {
                int v = m->term[e]; double n = (double)m->count[e];
                const double* th = ctm_theta(m, mod, e);
                for (int k = 0; k < m->K[mod]; ++k) {
                    double nt = th[k] * n;
                    if (!m->n_feat) m->gamma[go + (size_t)k * m->V[mod] + v] += nt;
                    else {
                        int SJ = ctm_SJ(m, mod); size_t fo = ctm_foff(m, mod);
                        for (int i = 0; i < m->n_feat[mod]; ++i) {
                            int f = m->features[fo + (size_t)i * m->V[mod] + v];
                            m->gamma[go + (size_t)k * SJ + ctm_joff(m, mod, i) + f] += nt;
                        }
                    }
                }
            }
        }
    }
    orc_ctm_update_Elnphi(m);
}

/* MMCTM.jl:145-154 */
void orc_ctm_update_props(orc_ctm* m)
{
    for (int d = 0; d < m->D; ++d) {
        int off = 0;
        for (int mod = 0; mod < m->M; ++mod) {
            const double* eta = m->lambda + (size_t)m->MK * d + off; double* p = m->props + (size_t)m->MK * d + off;
            double s = 0.0;
            for (int k = 0; k < m->K[mod]; ++k) s += exp(eta[k]);
            for (int k = 0; k < m->K[mod]; ++k) p[k] = exp(eta[k]) / s;
            off += m->K[mod];
        }
    }
}

/* MMCTM.jl:244-250 */
void orc_ctm_update_phi(orc_ctm* m)
{
    for (int mod = 0; mod < m->M; ++mod) {
        size_t go = ctm_goff(m, mod); int V = m->V[mod];
        for (int k = 0; k < m->K[mod]; ++k) {
            double s = 0.0; for (int v = 0; v < V; ++v) s += m->gamma[go + (size_t)k * V + v];
            for (int v = 0; v < V; ++v) m->phi[go + (size_t)k * V + v] = m->gamma[go + (size_t)k * V + v] / s;
        }
    }
}

/* MMCTM.jl:384-448 (props/phi stored) ; IMMCTM.jl:362-428 (props and phi recomputed from lambda, gamma) */
void orc_ctm_loglik(const orc_ctm* m, double* ll)
{
    int off = 0;
    for (int mod = 0; mod < m->M; ++mod) {
        const int64_t* dp = m->doc_ptr + (size_t)mod * (m->D + 1);
        int Km = m->K[mod]; size_t go = ctm_goff(m, mod);
        double* phin = NULL; int SJ = 0; size_t fo = 0;
        if (m->n_feat) { /* IMMCTM.jl:417-420: phi[k][i] = gamma[k][i] ./ sum(gamma[k][i]) */
            SJ = ctm_SJ(m, mod); fo = ctm_foff(m, mod); int ao = ctm_aoff(m, mod);
            phin = (double*)malloc(sizeof(double) * (size_t)Km * SJ);
            for (int k = 0; k < Km; ++k) { int jo = 0;
                for (int i = 0; i < m->n_feat[mod]; ++i) { double s = 0.0; int Ji = m->J[ao + i];
                    for (int j = 0; j < Ji; ++j) s += m->gamma[go + (size_t)k * SJ + jo + j];
                    for (int j = 0; j < Ji; ++j) phin[(size_t)k * SJ + jo + j] = m->gamma[go + (size_t)k * SJ + jo + j] / s;
                    jo += Ji; } }
        }
        double* pr = (double*)malloc(sizeof(double) * Km);
        double tot = 0.0; int64_t N = 0;
        for (int d = 0; d < m->D; ++d) {
            int64_t docN = 0; for (int64_t e = dp[d]; e < dp[d + 1]; ++e) docN += m->count[e];
            if (docN <= 0) continue;
            if (m->n_feat) { const double* eta = m->lambda + (size_t)m->MK * d + off; double s = 0.0;
                for (int k = 0; k < Km; ++k) s += exp(eta[k]);
                for (int k = 0; k < Km; ++k) pr[k] = exp(eta[k]) / s;
            } else for (int k = 0; k < Km; ++k) pr[k] = m->props[(size_t)m->MK * d + off + k];
            double dl = 0.0;
            for (int64_t e = dp[d]; e < dp[d + 1]; ++e) {
                int v = m->term[e]; double pw = 0.0;
                for (int k = 0; k < Km; ++k) {
                    if (!m->n_feat) pw += pr[k] * m->phi[go + (size_t)k * m->V[mod] + v];
                    else { double t = pr[k];
                        for (int i = 0; i < m->n_feat[mod]; ++i) t *= phin[(size_t)k * SJ + ctm_joff(m, mod, i) + m->features[fo + (size_t)i * m->V[mod] + v]];
                        pw += t; }
                }
                dl += (double)m->count[e] * log(pw);
            }
            double doc_ll = dl / (double)docN;           /* MMCTM.jl:399 */
            tot += doc_ll * (double)docN; N += docN;     /* MMCTM.jl:412-413 */
        }
        ll[mod] = tot / (double)N;
        free(pr); free(phin);
        off += Km;
    }
}

/* MMCTM.jl:271-382 / IMMCTM.jl:247-360 */
double orc_ctm_elbo(const orc_ctm* m, double* terms)
{
    double t[7] = {0, 0, 0, 0, 0, 0, 0};
    int n = m->MK;
    /* ElnPphi :271-284 ; ElnQphi :338-350 */
    for (int mod = 0; mod < m->M; ++mod) {
        size_t go = ctm_goff(m, mod);
        if (!m->n_feat) {
            int V = m->V[mod];
            double* fillv = (double*)malloc(sizeof(double) * V);
            for (int v = 0; v < V; ++v) fillv[v] = m->alpha[mod];
            for (int k = 0; k < m->K[mod]; ++k) {
                t[0] -= orc_logmvbeta(V, fillv);
                for (int v = 0; v < V; ++v) t[0] += (m->alpha[mod] - 1) * m->Elnphi[go + (size_t)k * V + v];
                t[4] += -orc_logmvbeta(V, m->gamma + go + (size_t)k * V);
                for (int v = 0; v < V; ++v) t[4] += (m->gamma[go + (size_t)k * V + v] - 1) * m->Elnphi[go + (size_t)k * V + v];
            }
            free(fillv);
        } else {
            int SJ = ctm_SJ(m, mod), ao = ctm_aoff(m, mod);
            for (int k = 0; k < m->K[mod]; ++k) { int jo = 0;
                for (int i = 0; i < m->n_feat[mod]; ++i) { int Ji = m->J[ao + i];
                    double* fillv = (double*)malloc(sizeof(double) * Ji);
                    for (int j = 0; j < Ji; ++j) fillv[j] = m->alpha[ao + i];
                    t[0] -= orc_logmvbeta(Ji, fillv);
                    for (int j = 0; j < Ji; ++j) t[0] += (m->alpha[ao + i] - 1) * m->Elnphi[go + (size_t)k * SJ + jo + j];
                    t[4] += -orc_logmvbeta(Ji, m->gamma + go + (size_t)k * SJ + jo);
                    for (int j = 0; j < Ji; ++j) t[4] += (m->gamma[go + (size_t)k * SJ + jo + j] - 1) * m->Elnphi[go + (size_t)k * SJ + jo + j];
                    free(fillv); jo += Ji; } }
        }
    }
    /* ElnPeta :286-300 */
    double logdet; int sg;
    orc_inv_logdet(n, m->invSigma, NULL, &logdet, &sg);
    double* buf = (double*)malloc(sizeof(double) * (size_t)n * 3);
    double *diff = buf, *st = buf + n, *ndz = buf + 2 * n;
    for (int d = 0; d < m->D; ++d) {
        const double* lam = m->lambda + (size_t)n * d; const double* nu = m->nu + (size_t)n * d;
        double tr = 0.0, quad = 0.0;
        for (int i = 0; i < n; ++i) { diff[i] = lam[i] - m->mu[i]; tr += nu[i] * m->invSigma[i + (size_t)n * i]; }
        for (int i = 0; i < n; ++i) { double s = 0.0; for (int j = 0; j < n; ++j) s += m->invSigma[i + (size_t)n * j] * diff[j]; quad += diff[i] * s; }
        t[1] += 0.5 * (logdet - n * log(2 * M_PI) - tr - quad);
        /* ElnPZ :302-316 */
        orc_ctm_calc_sumtheta(m, d, st); orc_ctm_calc_Ndivzeta(m, d, ndz);
        double a = 0.0, b = 0.0, sN = 0.0, c = 0.0;
        for (int i = 0; i < n; ++i) { a += lam[i] * st[i]; b += ndz[i] * exp(lam[i] + 0.5 * nu[i]); }
        for (int mod = 0; mod < m->M; ++mod) { double Ndm = (double)ctm_N(m, mod, d); sN += Ndm; c += Ndm * log(m->zeta[mod + (size_t)m->M * d]); }
        t[2] += a; t[2] -= b - sN; t[2] -= c;
        /* ElnQeta :352-358 */
        double sl = 0.0; for (int i = 0; i < n; ++i) sl += log(nu[i]);
        t[5] += -0.5 * (sl + n * (log(2 * M_PI) + 1));
        /* ElnPX :318-336 ; ElnQZ :360-370 */
        for (int mod = 0; mod < m->M; ++mod) {
            const int64_t* dp = m->doc_ptr + (size_t)mod * (m->D + 1); size_t go = ctm_goff(m, mod);
            for (int64_t e = dp[d]; e < dp[d + 1]; ++e) {
                int v = m->term[e]; double cnt = (double)m->count[e]; const double* th = ctm_theta(m, mod, e);
                for (int k = 0; k < m->K[mod]; ++k) {
                    if (!m->n_feat) t[3] += cnt * th[k] * m->Elnphi[go + (size_t)k * m->V[mod] + v];
                    else { int SJ = ctm_SJ(m, mod); size_t fo = ctm_foff(m, mod);
                        for (int i = 0; i < m->n_feat[mod]; ++i)
                            t[3] += cnt * th[k] * m->Elnphi[go + (size_t)k * SJ + ctm_joff(m, mod, i) + m->features[fo + (size_t)i * m->V[mod] + v]]; }
                    t[6] += cnt * log(pow(th[k], th[k]));
                }
            }
        }
    }
    free(buf);
    if (terms) memcpy(terms, t, sizeof t);
    return t[0] + t[1] + t[2] + t[3] - t[4] - t[5] - t[6];
}

/* α_objective  common.jl:38-46 (value returned, gradient written when grad != NULL; the objective is MAXIMISED) */
double orc_alpha_objective(double alpha, double* grad, double sum_Elnphi, int K, int V)
{
    if (grad) *grad = (double)K * V * (orc_digamma(V * alpha) - orc_digamma(alpha)) + sum_Elnphi;
    return K * (orc_lgamma(V * alpha) - V * orc_lgamma(alpha)) + alpha * sum_Elnphi;
}

typedef struct { double s; int K, V; } alpha_ctx;
static double cb_alpha(int n, const double* x, double* grad, void* p)
{
    const alpha_ctx* c = (const alpha_ctx*)p; double g;
    double v = orc_alpha_objective(x[0], &g, c->s, c->K, c->V);
    if (grad) grad[0] = -g;
    return -v;
}

/* update_α!  MMCTM.jl:252-269 / IMMCTM.jl:225-244: 1-D LD_MMA, lower bound 1e-7, xtol_rel = xtol_abs = 1e-5, start = alpha */
void orc_ctm_update_alpha(orc_ctm* m)
{
    const double lb = 1e-7;
    for (int mod = 0; mod < m->M; ++mod) {
        size_t go = ctm_goff(m, mod); int Km = m->K[mod];
        if (!m->n_feat) {
            int V = m->V[mod];
            /* sum(sum(Elnϕ[m][k] for k in 1:K)): the K vectors are added elementwise, then summed over v */
            double s = 0.0;
            for (int v = 0; v < V; ++v) { double c = 0.0; for (int k = 0; k < Km; ++k) c += m->Elnphi[go + (size_t)k * V + v]; s += c; }
            alpha_ctx c = { s, Km, V };
            double x = m->alpha[mod];
            orc_mma_minimize(1, cb_alpha, &c, &lb, NULL, &x, NULL, 1e-5, 1e-5, m->xtol_rule, m->max_eval, NULL);
            m->alpha[mod] = x;
        } else {
            int SJ = ctm_SJ(m, mod), ao = ctm_aoff(m, mod), jo = 0;
            for (int i = 0; i < m->n_feat[mod]; ++i) {
                int Ji = m->J[ao + i]; double s = 0.0;
                for (int j = 0; j < Ji; ++j) { double c = 0.0; for (int k = 0; k < Km; ++k) c += m->Elnphi[go + (size_t)k * SJ + jo + j]; s += c; }
                alpha_ctx c = { s, Km, Ji };
                double x = m->alpha[ao + i];
                orc_mma_minimize(1, cb_alpha, &c, &lb, NULL, &x, NULL, 1e-5, 1e-5, m->xtol_rule, m->max_eval, NULL);
                m->alpha[ao + i] = x;
                jo += Ji;
            }
        }
    }
}

/* constructor state: MMCTM.jl:44-86 / IMMCTM.jl:47-73 (gamma must already hold the random init) */
void orc_ctm_init(orc_ctm* m)
{
    int n = m->MK;
    for (int i = 0; i < n; ++i) m->mu[i] = 0.0;
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) m->Sigma[i + (size_t)n * j] = m->invSigma[i + (size_t)n * j] = (i == j) ? 1.0 : 0.0;
    for (int mod = 0; mod < m->M; ++mod) {
        const int64_t* dp = m->doc_ptr + (size_t)mod * (m->D + 1);
        for (int64_t e = dp[0]; e < dp[m->D]; ++e) { double* th = ctm_theta(m, mod, e); for (int k = 0; k < m->K[mod]; ++k) th[k] = 1.0 / m->K[mod]; }
    }
    if (m->arith) orc_twin_topics(m, NULL); else orc_ctm_update_Elnphi(m);
    if (m->phi) { size_t tot = ctm_goff(m, m->M); for (size_t i = 0; i < tot; ++i) m->phi[i] = m->gamma[i]; } /* MMCTM.jl:80 deepcopy(gamma) */
    for (size_t i = 0; i < (size_t)n * m->D; ++i) { m->lambda[i] = 0.0; m->nu[i] = 1.0; }
    for (int d = 0; d < m->D; ++d) orc_ctm_update_zeta(m, d);
}

/* MMCTM.jl:457-494 / IMMCTM.jl:437-466 */
int orc_ctm_fit(orc_ctm* m, int maxiter, double tol, int update_sigma, int auto_alpha, double* ll_hist, int* n_iter,
                int* converged, double* elbo)
{
    *converged = 0; int it = 0;
    for (int iter = 1; iter <= maxiter; ++iter) {
        if (m->arith) orc_twin_pass(m, update_sigma || m->n_feat);      /* the same steps in device order (mmm_twin.c) */
        else {
            for (int d = 0; d < m->D; ++d) orc_ctm_fitdoc(m, d);
            orc_ctm_update_mu(m);
            if (update_sigma || m->n_feat) orc_ctm_update_Sigma(m);
            orc_ctm_update_gamma(m);
        }
        if (auto_alpha) orc_ctm_update_alpha(m);                 /* MMCTM.jl:472-474 / IMMCTM.jl:448-450 */
        if (!m->n_feat) { orc_ctm_update_props(m); orc_ctm_update_phi(m); }
        orc_ctm_loglik(m, ll_hist + (size_t)m->M * it); ++it;
        if (it > 10) { /* common.jl:48-51 */
            double rel = 0.0;
            for (int mod = 0; mod < m->M; ++mod) {
                double a = ll_hist[(size_t)m->M * (it - 2) + mod], b = ll_hist[(size_t)m->M * (it - 1) + mod];
                double r = fabs(a - b) / fabs(b); if (r > rel) rel = r;
            }
            if (rel < tol) { *converged = 1; break; }
        }
    }
    *n_iter = it;
    if (elbo) *elbo = orc_ctm_elbo(m, NULL);
    return 0;
}

/* unsmoothed_update_θ!  MMCTM.jl:496-509: theta[k, w] = exp(lambda[k]) * phi[m][k][v], normalised per term */
void orc_ctm_unsmoothed_update_theta(orc_ctm* m, int d)
{
    const double* lam = m->lambda + (size_t)m->MK * d;
    int offset = 0;
    for (int mod = 0; mod < m->M; ++mod) {
        const int64_t* dp = m->doc_ptr + (size_t)mod * (m->D + 1);
        int Km = m->K[mod];
        size_t go = ctm_goff(m, mod);
        for (int64_t e = dp[d]; e < dp[d + 1]; ++e) {
            int v = m->term[e];
            double* th = ctm_theta(m, mod, e);
            double s = 0.0;
            for (int k = 0; k < Km; ++k) { th[k] = exp(lam[offset + k]) * m->phi[go + (size_t)k * m->V[mod] + v]; s += th[k]; }
            for (int k = 0; k < Km; ++k) th[k] /= s;
        }
        offset += Km;
    }
}

/* The loops of transform (MMCTM.jl:521-549; flags & 1: unsmoothed theta, flags & 2: fit_gaussian) and of fit_heldout /
 * predict_modality_η (MMCTM.jl:565-583, :605-621; IMMCTM.jl:479-495, :516-532; flags = 0) on a constructor-initialised
 * model whose globals were copied in by the caller.  props are refreshed every pass for MMCTM (transform and fit_heldout
 * do; MMCTM's predict_modality_η reads them uninitialised -- the IMMCTM twin recomputes them, which is what is done here). */
int orc_ctm_infer(orc_ctm* m, int flags, int maxiter, double tol, double* ll_hist, int* n_iter, int* converged)
{
    *converged = 0; int it = 0;
    for (int iter = 1; iter <= maxiter; ++iter) {
        if (m->arith) {                                   /* the same steps in device order (mmm_twin.c) */
            if (iter == 1) orc_twin_tables_from_Elnphi(m);
            orc_twin_infer_pass(m, flags);
        } else {
            for (int d = 0; d < m->D; ++d) {
                orc_ctm_update_zeta(m, d);
                if (flags & 1) orc_ctm_unsmoothed_update_theta(m, d); else orc_ctm_update_theta(m, d);
                orc_ctm_update_nu(m, d);
                orc_ctm_update_lambda(m, d);
            }
            if (flags & 2) { orc_ctm_update_mu(m); orc_ctm_update_Sigma(m); }
        }
        if (!m->n_feat) orc_ctm_update_props(m);
        orc_ctm_loglik(m, ll_hist + (size_t)m->M * it); ++it;
        if (it > 10) {
            double rel = 0.0;
            for (int mod = 0; mod < m->M; ++mod) {
                double a = ll_hist[(size_t)m->M * (it - 2) + mod], b = ll_hist[(size_t)m->M * (it - 1) + mod];
                double r = fabs(a - b) / fabs(b); if (r > rel) rel = r;
            }
            if (rel < tol) { *converged = 1; break; }
        }
    }
    *n_iter = it;
    return 0;
}
