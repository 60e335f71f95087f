/*
 * rc_oracle.c — CPU restatement of RedClust.jl's Gibbs-sweep hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the timed CPU baseline.  The HIP product path (redclust.jl_amd/csrc) never calls it.
 *
 * PARITY STATUS: "parity unpinned" against the Julia package itself — the reference's own tests hold
 * no golden vector / known-answer for the sampler, loglik, logprior or the co-clustering matrix
 *  and Julia is not installed in the build container.  What pins this file instead:
 * an independent NumPy/SciPy transcription (tests/np_transcription.py) on the three paper datasets of
 * the reference's data/example_datasets.h5, committed under tests/golden/ (npz files), plus the value tests the
 * reference does hold at this boundary (matsum/vecsum ≈ sum, adjacencymatrix / sortlabels structure,
 * test/test_utils.jl:10-40), restated in tests/.
 *
 * Two arithmetic modes of the SAME algorithm:
 *   literal — the reference's formulas as written (f64 sums in ascending member order, loggamma/log of
 *             the full-magnitude arguments, L2_i added to every candidate, minimum subtracted).
 *             This is the restatement proper.  cost_mode=1 additionally reproduces the reference's
 *             per-(point,cluster) member scans and its three strided gathers (mcmc.jl:195-214), for the
 *             CPU timing baseline ("faithful-cost"); cost_mode=0 buckets row i once ("single-pass").
 *             Both produce bit-identical results (same summation order).
 *   stable  — same algorithm, regrouped arithmetic: row sums accumulated exactly in
 *             64-bit fixed point (order-independent), lgamma differences tabulated per cluster size in
 *             long double, log(β+S) written as log β + log1p(S/β), the candidate-independent terms
 *             (L2_i and the subtracted minimum) dropped — they shift every candidate equally and do not
 *             change the Gumbel-max draw.  This is the arithmetic the HIP kernels implement; tests
 *             check stable == HIP exactly on labels/sizes and stable ≈ literal within tolerance.
 *
 * Reference citations are path:line under /root/reference (RedClust.jl v1.2.2).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double delta1, delta2, alpha, beta, zeta, gamma; /* src/types.jl:94-99 */
    double eta, sigma, u, v;                         /* src/types.jl:100-104 */
    int32_t repulsion;                               /* src/types.jl:106 */
    int64_t maxK;                                    /* src/types.jl:107 */
} orc_params;

/* ------------------------------------------------------------------------------------------------
 * Uniform source.  Julia's task-local RNG cannot be reproduced outside Julia, so
 * "identical RNG seeds" is defined at the uniform-stream level: the m uniforms that
 * sample_logweights (src/utils.jl:4) draws for point i in sweep t are u(seed, t, i, pos), pos =
 * 0..m-1 in candidate order, from Philox4x32-10 (Salmon et al., SC'11) keyed by the seed with the
 * counter (pos, i, t_lo, t_hi).  Strictly inside (0,1).
 * ---------------------------------------------------------------------------------------------- */
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    philox4x32_10(c, key[0], key[1]);
    memcpy(out, c, sizeof(c));
}

double orc_uniform(uint64_t seed, uint64_t sweep, uint64_t i, uint64_t pos)
{
    uint32_t c[4] = {(uint32_t)pos, (uint32_t)i, (uint32_t)sweep, (uint32_t)(sweep >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint64_t bits = (((uint64_t)c[0] << 32) | c[1]) >> 12; /* 52 bits */
    return ((double)bits + 0.5) * 0x1p-52;
}

/* ------------------------------------------------------------------------------------------------
 * MCMCData (src/types.jl:145-157): logD = log.(D - Diagonal(D) + I): off-diagonal log(D), diagonal 0.
 * Returns 0, or -1 if D is not symmetric (types.jl:149-151).
 * ---------------------------------------------------------------------------------------------- */
int orc_make_logD(int64_t n, const double *D, double *logD)
{
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j)
            if (D[i * n + j] != D[j * n + i]) return -1;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j)
            logD[i * n + j] = (i == j) ? 0.0 : log(D[i * n + j]);
    return 0;
}

/* MCMCState (src/types.jl:131-137): clustsizes = counts(clusts, 1:n), K = sum(clustsizes .> 0). */
int orc_state_from_labels(int64_t n, const int64_t *clusts, int64_t *clustsizes, int64_t *K)
{
    memset(clustsizes, 0, (size_t)n * sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) {
        if (clusts[i] < 1 || clusts[i] > n) return -1;
        clustsizes[clusts[i] - 1]++;
    }
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i) k += clustsizes[i] > 0;
    *K = k;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Fixed-point quantisation used by the stable mode (and by the HIP path): q = rint(x * 2^e) with
 * e = 62 - (exponent of max|x|) - ceil(log2 n), so that any sum of n entries fits in int64.
 * ---------------------------------------------------------------------------------------------- */
static int ceil_log2_i64(int64_t n)
{
    int b = 0;
    while (((int64_t)1 << b) < n) ++b;
    return b;
}

int orc_quant_exponent(int64_t n, const double *x, int64_t count)
{
    double m = 0.0;
    for (int64_t t = 0; t < count; ++t) {
        double a = fabs(x[t]);
        if (!(a <= 1.79769313486231570e308)) return -10000; /* non-finite */
        if (a > m) m = a;
    }
    if (m == 0.0) return 0;
    int ex;
    frexp(m, &ex); /* m = f * 2^ex, f in [0.5,1) => m < 2^ex */
    return 62 - ex - ceil_log2_i64(n);
}

/* 32-bit storage variant of the HIP path: every entry fits int32 (|q| < 2^30), sums are 64-bit. */
int orc_quant_exponent32(const double *x, int64_t count)
{
    double m = 0.0;
    for (int64_t t = 0; t < count; ++t) {
        double a = fabs(x[t]);
        if (!(a <= 1.79769313486231570e308)) return -10000;
        if (a > m) m = a;
    }
    if (m == 0.0) return 0;
    int ex;
    frexp(m, &ex);
    return 30 - ex;
}

void orc_quantize(const double *x, int64_t count, int e, int64_t *q)
{
    for (int64_t t = 0; t < count; ++t) q[t] = llrint(ldexp(x[t], e));
}

/* ------------------------------------------------------------------------------------------------
 * sample_logweights (src/utils.jl:2-6), literal: lp .-= minimum(lp); u = rand(m);
 * argmax(-log.(-log.(u)) .+ lp), first index wins ties.  Returns 0-based position.
 * ---------------------------------------------------------------------------------------------- */
static int64_t draw_literal(double *lp, int64_t m, uint64_t seed, uint64_t sweep, uint64_t i)
{
    double mn = lp[0];
    for (int64_t k = 1; k < m; ++k) if (lp[k] < mn) mn = lp[k];
    int64_t best = 0;
    double bestv = 0;
    for (int64_t k = 0; k < m; ++k) {
        lp[k] -= mn;
        double u = orc_uniform(seed, sweep, i, (uint64_t)k);
        double v = -log(-log(u)) + lp[k];
        if (k == 0 || v > bestv) { bestv = v; best = k; }
    }
    return best;
}

/* Workspace for the literal sweep: the six length-n label-indexed scratch vectors of
 * mcmc.jl:180-185 plus candidate lists. */
typedef struct {
    double *alpha_i, *beta_i, *zeta_i, *gamma_i, *sum_logD_i, *L2_ik_prime;
    double *sumD; /* single-pass bucket for D */
    int64_t *C_i, *cand, *members;
    double *L1, *L2, *lpr, *logprobs;
} lit_ws;

static int lit_ws_alloc(lit_ws *w, int64_t n)
{
    size_t d = (size_t)n * sizeof(double), l = (size_t)(n + 1) * sizeof(int64_t);
    w->alpha_i = calloc(1, d); w->beta_i = calloc(1, d); w->zeta_i = calloc(1, d);
    w->gamma_i = calloc(1, d); w->sum_logD_i = calloc(1, d); w->L2_ik_prime = calloc(1, d);
    w->sumD = calloc(1, d);
    w->C_i = malloc(l); w->cand = malloc(l); w->members = malloc(l);
    w->L1 = malloc(d + 8); w->L2 = malloc(d + 8); w->lpr = malloc(d + 8); w->logprobs = malloc(d + 8);
    return (w->alpha_i && w->beta_i && w->zeta_i && w->gamma_i && w->sum_logD_i && w->L2_ik_prime &&
            w->sumD && w->C_i && w->cand && w->members && w->L1 && w->L2 && w->lpr && w->logprobs) ? 0 : -1;
}
static void lit_ws_free(lit_ws *w)
{
    free(w->alpha_i); free(w->beta_i); free(w->zeta_i); free(w->gamma_i); free(w->sum_logD_i);
    free(w->L2_ik_prime); free(w->sumD); free(w->C_i); free(w->cand); free(w->members);
    free(w->L1); free(w->L2); free(w->lpr); free(w->logprobs);
}

/* One point of sample_labels_Gibbs! (src/mcmc.jl:192-252), literal arithmetic.  Labels 1-based.
 * On entry point i (0-based) has ALREADY been removed (clustsizes decremented, clusts[i] = -1).
 * Fills w->cand[0..m), w->logprobs[0..m) (before the minimum is subtracted) and returns m. */
static int64_t lit_point_scores(int64_t n, const double *D, const double *logD, const int64_t *clusts,
                                const int64_t *clustsizes, const orc_params *P, double r, double p,
                                int64_t i, int cost_mode, lit_ws *w)
{
    const double d1 = P->delta1, d2 = P->delta2, al = P->alpha, be = P->beta, ze = P->zeta, ga = P->gamma;
    const double abratio = al * log(be) - lgamma(al);          /* mcmc.jl:186 */
    const double zgratio = ze * log(ga) - lgamma(ze);          /* mcmc.jl:187 */
    const double lg_d1 = lgamma(d1), lg_d2 = lgamma(d2);       /* mcmc.jl:188-189 */
    const double logp = log(p), log1mp = log(1 - p);           /* mcmc.jl:190-191 */
    const double rep = P->repulsion ? 1.0 : 0.0;

    /* C_i = findall(clustsizes .> 0)  (mcmc.jl:195) */
    int64_t K_i = 0;
    for (int64_t k = 0; k < n; ++k) if (clustsizes[k] > 0) w->C_i[K_i++] = k + 1;
    int64_t m = K_i;
    memcpy(w->cand, w->C_i, (size_t)K_i * sizeof(int64_t));
    if ((P->maxK == 0 || K_i < P->maxK) && K_i < n) {          /* mcmc.jl:198-199 */
        int64_t e = 0;
        while (clustsizes[e] != 0) ++e;                        /* findfirst(clustsizes .== 0) */
        w->cand[m++] = e + 1;
    }

    if (cost_mode == 0) {
        /* single-pass: bucket row i by label; per-cluster order = ascending j (same as the scan) */
        for (int64_t t = 0; t < K_i; ++t) { w->sumD[w->C_i[t] - 1] = 0; w->sum_logD_i[w->C_i[t] - 1] = 0; }
        const double *Di = D + i * n, *Li = logD + i * n;
        for (int64_t j = 0; j < n; ++j) {
            int64_t c = clusts[j];
            if (c < 1) continue;
            w->sumD[c - 1] += Di[j];
            w->sum_logD_i[c - 1] += Li[j];
        }
        for (int64_t t = 0; t < K_i; ++t) {
            int64_t k = w->C_i[t] - 1;
            double sz = (double)clustsizes[k];
            w->alpha_i[k] = al + d1 * sz;                      /* mcmc.jl:209 */
            w->beta_i[k] = be + w->sumD[k];                    /* mcmc.jl:210 */
            w->zeta_i[k] = ze + d2 * sz;                       /* mcmc.jl:211 */
            w->gamma_i[k] = ga + w->sumD[k];                   /* mcmc.jl:212 */
        }
    } else {
        /* faithful-cost: findall(clusts .== k) per cluster, then matsum(D,[i],clust_k) twice and
         * matsum(logD,[i],clust_k) once, each a stride-n gather x[i, j] of a column-major matrix
         * (mcmc.jl:206-214, utils.jl:9-17). */
        for (int64_t t = 0; t < K_i; ++t) {
            int64_t lab = w->C_i[t], k = lab - 1, nm = 0;
            int64_t *mem = malloc((size_t)(n + 1) * sizeof(int64_t)); /* findall allocates */
            for (int64_t j = 0; j < n; ++j) if (clusts[j] == lab) mem[nm++] = j;
            double sz = (double)clustsizes[k];
            double s1 = 0, s2 = 0, s3 = 0;
            for (int64_t q = 0; q < nm; ++q) s1 += D[i + n * mem[q]];
            for (int64_t q = 0; q < nm; ++q) s2 += D[i + n * mem[q]];
            for (int64_t q = 0; q < nm; ++q) s3 += logD[i + n * mem[q]];
            free(mem);
            w->alpha_i[k] = al + d1 * sz;
            w->beta_i[k] = be + s1;
            w->zeta_i[k] = ze + d2 * sz;
            w->gamma_i[k] = ga + s2;
            w->sum_logD_i[k] = s3;
        }
    }

    /* cohesion + prior ratio (mcmc.jl:221-237) */
    for (int64_t k = 0; k < m; ++k) {
        int64_t c = w->cand[k] - 1;
        int64_t szc = clustsizes[c];
        if (szc == 0) { /* new cluster (only possible for the last slot) */
            w->lpr[k] = log((double)(K_i + 1)) + r * log1mp;   /* mcmc.jl:229 */
            w->L1[k] = 0;
        } else {
            double sz = (double)szc;
            w->L1[k] = lgamma(w->alpha_i[c]) + abratio - w->alpha_i[c] * log(w->beta_i[c]) +
                       (d1 - 1) * w->sum_logD_i[c] - sz * lg_d1;                 /* mcmc.jl:223-225 */
            w->lpr[k] = log(sz + 1) + logp + log(sz - 1 + r) - log(sz);          /* mcmc.jl:226 */
        }
    }
    /* repulsion (mcmc.jl:239-246) */
    double L2_i = 0;
    for (int64_t t = 0; t < K_i; ++t) {
        int64_t c = w->C_i[t] - 1;
        w->L2_ik_prime[c] = lgamma(w->zeta_i[c]) - w->zeta_i[c] * log(w->gamma_i[c]) + zgratio +
                            (d2 - 1) * w->sum_logD_i[c] - (double)clustsizes[c] * lg_d2;
    }
    for (int64_t t = 0; t < K_i; ++t) L2_i += w->L2_ik_prime[w->C_i[t] - 1];      /* vecsum, :243 */
    for (int64_t k = 0; k < m; ++k) {
        int64_t c = w->cand[k] - 1;
        /* Julia Bool multiplier is a strong zero (mcmc.jl:245): stale entries never leak */
        w->L2[k] = (clustsizes[c] != 0) ? (L2_i - w->L2_ik_prime[c]) : L2_i;
    }
    for (int64_t k = 0; k < m; ++k)                                               /* mcmc.jl:247 */
        w->logprobs[k] = w->lpr[k] + (w->L1[k] + (rep != 0.0 ? w->L2[k] : 0.0));
    return m;
}

/* sample_labels_Gibbs! (src/mcmc.jl:158-256), literal.  clusts 1-based, clustsizes length n.
 * cost_mode: 0 single-pass, 1 faithful-cost.  Returns 0 or <0 on allocation failure.
 * The _range form visits points i_begin..i_end-1 only (bounded CPU-baseline sample in bench.py). */
int orc_sweep_literal_range(int64_t n, const double *D, const double *logD, int64_t *clusts,
                            int64_t *clustsizes, int64_t *K, const orc_params *P, double r, double p,
                            uint64_t seed, uint64_t sweep, int cost_mode, int64_t i_begin, int64_t i_end)
{
    lit_ws w;
    if (lit_ws_alloc(&w, n)) { lit_ws_free(&w); return -2; }
    for (int64_t i = i_begin; i < i_end; ++i) {
        clustsizes[clusts[i] - 1] -= 1;                         /* mcmc.jl:193 */
        clusts[i] = -1;                                         /* mcmc.jl:194 */
        int64_t m = lit_point_scores(n, D, logD, clusts, clustsizes, P, r, p, i, cost_mode, &w);
        int64_t k = draw_literal(w.logprobs, m, seed, sweep, (uint64_t)i);        /* mcmc.jl:249 */
        int64_t ci_new = w.cand[k];
        clusts[i] = ci_new;                                     /* mcmc.jl:251 */
        clustsizes[ci_new - 1] += 1;                            /* mcmc.jl:252 */
    }
    int64_t k = 0;
    for (int64_t t = 0; t < n; ++t) k += clustsizes[t] > 0;     /* mcmc.jl:254 */
    *K = k;
    lit_ws_free(&w);
    return 0;
}

int orc_sweep_literal(int64_t n, const double *D, const double *logD, int64_t *clusts,
                      int64_t *clustsizes, int64_t *K, const orc_params *P, double r, double p,
                      uint64_t seed, uint64_t sweep, int cost_mode)
{
    return orc_sweep_literal_range(n, D, logD, clusts, clustsizes, K, P, r, p, seed, sweep, cost_mode, 0, n);
}

/* Candidate labels and log-weights (mcmc.jl:247, before sample_logweights) for ONE point of the
 * CURRENT state, without modifying the state.  out arrays need room for n+1 entries. */
int64_t orc_point_scores_literal(int64_t n, const double *D, const double *logD, const int64_t *clusts,
                                 const int64_t *clustsizes, const orc_params *P, double r, double p,
                                 int64_t i, int64_t *out_cand, double *out_logprobs)
{
    lit_ws w;
    if (lit_ws_alloc(&w, n)) { lit_ws_free(&w); return -2; }
    int64_t *c2 = malloc((size_t)n * sizeof(int64_t)), *s2 = malloc((size_t)n * sizeof(int64_t));
    memcpy(c2, clusts, (size_t)n * sizeof(int64_t));
    memcpy(s2, clustsizes, (size_t)n * sizeof(int64_t));
    s2[c2[i] - 1] -= 1; c2[i] = -1;
    int64_t m = lit_point_scores(n, D, logD, c2, s2, P, r, p, i, 0, &w);
    memcpy(out_cand, w.cand, (size_t)m * sizeof(int64_t));
    memcpy(out_logprobs, w.logprobs, (size_t)m * sizeof(double));
    free(c2); free(s2);
    lit_ws_free(&w);
    return m;
}

/* ------------------------------------------------------------------------------------------------
 * Stable mode.
 * Size table A[s], s = 1..n (A[0] unused):
 *   A[s] = [lgΓ(α+δ1 s) − lgΓ(α) − δ1 s log β − s lgΓ(δ1)]
 *        − rep·[lgΓ(ζ+δ2 s) − lgΓ(ζ) − δ2 s log γ − s lgΓ(δ2)] + log((s+1)/s)
 * which collects every size-only term of L1 (mcmc.jl:223-225), −L2' (mcmc.jl:240-241) and the prior
 * ratio log(s+1) − log(s) (mcmc.jl:226), evaluated in long double.
 * ---------------------------------------------------------------------------------------------- */
void orc_size_table(int64_t n, const orc_params *P, double *A)
{
    long double d1 = P->delta1, d2 = P->delta2, al = P->alpha, be = P->beta, ze = P->zeta, ga = P->gamma;
    long double lga = lgammal(al), lgz = lgammal(ze), lgd1 = lgammal(d1), lgd2 = lgammal(d2);
    long double lb = logl(be), lg = logl(ga);
    A[0] = 0;
    for (int64_t s = 1; s <= n; ++s) {
        long double S = (long double)s;
        long double t1 = lgammal(al + d1 * S) - lga - d1 * S * lb - S * lgd1;
        long double t2 = lgammal(ze + d2 * S) - lgz - d2 * S * lg - S * lgd2;
        long double v = t1 - (P->repulsion ? t2 : 0.0L) + logl((S + 1) / S);
        A[s] = (double)v;
    }
}

/* Score of an existing-cluster candidate in stable arithmetic (everything but the Gumbel noise):
 * A[s] + log p + log(s−1+r) + cL·SL − (α+δ1 s)·log1p(SD/β) + rep·(ζ+δ2 s)·log1p(SD/γ).
 * Equals mcmc.jl:247's logprobs[k] − L2_i.  SD, SL are the real-valued row sums. */
static inline double stable_score(const orc_params *P, const double *A, int64_t s, double SD, double SL,
                                  double logp, double r)
{
    const double cL = (P->delta1 - 1) - (P->repulsion ? (P->delta2 - 1) : 0.0);
    double base = A[s] + (logp + log((double)s - 1 + r));
    double x1 = log1p(SD / P->beta);
    double lik = cL * SL - (P->alpha + P->delta1 * (double)s) * x1;
    if (P->repulsion) lik += (P->zeta + P->delta2 * (double)s) * log1p(SD / P->gamma);
    return base + lik;
}

/* sample_labels_Gibbs! in stable arithmetic.  Dq/Lq are the fixed-point matrices (row-major, symmetric),
 * eD/eL their exponents (value = q * 2^-e), A the size table.  Optional n_changes output. */
int orc_sweep_stable(int64_t n, const int64_t *Dq, const int64_t *Lq, int eD, int eL, const double *A,
                     int64_t *clusts, int64_t *clustsizes, int64_t *K, const orc_params *P, double r,
                     double p, uint64_t seed, uint64_t sweep, int64_t *n_changes)
{
    int64_t *sD = calloc((size_t)n, sizeof(int64_t)), *sL = calloc((size_t)n, sizeof(int64_t));
    int64_t *C_i = malloc((size_t)(n + 1) * sizeof(int64_t));
    if (!sD || !sL || !C_i) { free(sD); free(sL); free(C_i); return -2; }
    const double scD = ldexp(1.0, -eD), scL = ldexp(1.0, -eL);
    const double logp = log(p), log1mp = log(1 - p);
    int64_t changes = 0;
    for (int64_t i = 0; i < n; ++i) {
        int64_t old = clusts[i];
        clustsizes[old - 1] -= 1;
        clusts[i] = -1;
        int64_t K_i = 0;
        for (int64_t k = 0; k < n; ++k) if (clustsizes[k] > 0) { C_i[K_i++] = k + 1; sD[k] = 0; sL[k] = 0; }
        const int64_t *Di = Dq + i * n, *Li = Lq + i * n;
        for (int64_t j = 0; j < n; ++j) {
            int64_t c = clusts[j];
            if (c < 1) continue;
            sD[c - 1] += Di[j];
            sL[c - 1] += Li[j];
        }
        int64_t best = 0;
        double bestv = 0;
        int64_t m = K_i;
        for (int64_t t = 0; t < K_i; ++t) {
            int64_t c = C_i[t] - 1;
            double v = stable_score(P, A, clustsizes[c], (double)sD[c] * scD, (double)sL[c] * scL, logp, r);
            double u = orc_uniform(seed, sweep, (uint64_t)i, (uint64_t)t);
            v = v + (-log(-log(u)));
            if (t == 0 || v > bestv) { bestv = v; best = t; }
        }
        int64_t newlab = 0;
        if ((P->maxK == 0 || K_i < P->maxK) && K_i < n) {
            int64_t e = 0;
            while (clustsizes[e] != 0) ++e;
            newlab = e + 1;
            double v = log((double)(K_i + 1)) + r * log1mp;
            double u = orc_uniform(seed, sweep, (uint64_t)i, (uint64_t)K_i);
            v = v + (-log(-log(u)));
            if (K_i == 0 || v > bestv) { bestv = v; best = K_i; }
            m = K_i + 1;
        }
        (void)m;
        int64_t ci_new = (best < K_i) ? C_i[best] : newlab;
        clusts[i] = ci_new;
        clustsizes[ci_new - 1] += 1;
        changes += (ci_new != old);
    }
    int64_t k = 0;
    for (int64_t t = 0; t < n; ++t) k += clustsizes[t] > 0;
    *K = k;
    if (n_changes) *n_changes = changes;
    free(sD); free(sL); free(C_i);
    return 0;
}

/* Stable per-candidate scores (no Gumbel) for one point of the current state; for cross-checks. */
int64_t orc_point_scores_stable(int64_t n, const int64_t *Dq, const int64_t *Lq, int eD, int eL,
                                const double *A, const int64_t *clusts, const int64_t *clustsizes,
                                const orc_params *P, double r, double p, int64_t i, int64_t *out_cand,
                                double *out_scores)
{
    int64_t *sz = malloc((size_t)n * sizeof(int64_t));
    int64_t *sD = calloc((size_t)n, sizeof(int64_t)), *sL = calloc((size_t)n, sizeof(int64_t));
    memcpy(sz, clustsizes, (size_t)n * sizeof(int64_t));
    sz[clusts[i] - 1] -= 1;
    const double scD = ldexp(1.0, -eD), scL = ldexp(1.0, -eL);
    for (int64_t j = 0; j < n; ++j) {
        if (j == i) continue;
        sD[clusts[j] - 1] += Dq[i * n + j];
        sL[clusts[j] - 1] += Lq[i * n + j];
    }
    int64_t m = 0, K_i = 0;
    for (int64_t k = 0; k < n; ++k) if (sz[k] > 0) {
        out_cand[m] = k + 1;
        out_scores[m] = stable_score(P, A, sz[k], (double)sD[k] * scD, (double)sL[k] * scL, log(p), r);
        ++m; ++K_i;
    }
    if ((P->maxK == 0 || K_i < P->maxK) && K_i < n) {
        int64_t e = 0;
        while (sz[e] != 0) ++e;
        out_cand[m] = e + 1;
        out_scores[m] = log((double)(K_i + 1)) + r * log(1 - p);
        ++m;
    }
    free(sz); free(sD); free(sL);
    return m;
}

/* ------------------------------------------------------------------------------------------------
 * loglik (src/mcmc.jl:1-56), literal: per non-empty cluster full block sums (D's diagonal as stored,
 * logD's diagonal 0), per pair k<t cross-block sums.  Summation order: rows ascending, columns
 * ascending within the block (matsum's @turbo order is unspecified).
 * ---------------------------------------------------------------------------------------------- */
double orc_loglik_literal(int64_t n, const double *D, const double *logD, const int64_t *clusts,
                          const int64_t *clustsizes, const orc_params *P)
{
    const double d1 = P->delta1, d2 = P->delta2, al = P->alpha, be = P->beta, ze = P->zeta, ga = P->gamma;
    const double abratio = al * log(be) - lgamma(al), zgratio = ze * log(ga) - lgamma(ze);
    const double lg_d1 = lgamma(d1), lg_d2 = lgamma(d2);
    int64_t K = 0;
    int64_t *C = malloc((size_t)n * sizeof(int64_t));
    int64_t *slot = malloc((size_t)n * sizeof(int64_t)); /* label-1 -> index in C */
    for (int64_t k = 0; k < n; ++k) { slot[k] = -1; if (clustsizes[k] > 0) { slot[k] = K; C[K++] = k + 1; } }
    /* block sums BD[k][t] = Σ_{i∈k} Σ_{j∈t} D[i,j] accumulated row-by-row in ascending (i, j) */
    double *BD = calloc((size_t)(K * K), sizeof(double)), *BL = calloc((size_t)(K * K), sizeof(double));
    for (int64_t i = 0; i < n; ++i) {
        int64_t a = slot[clusts[i] - 1];
        for (int64_t j = 0; j < n; ++j) {
            int64_t b = slot[clusts[j] - 1];
            BD[a * K + b] += D[i * n + j];
            BL[a * K + b] += logD[i * n + j];
        }
    }
    double L1 = 0;
    for (int64_t k = 0; k < K; ++k) {                                   /* mcmc.jl:26-36 */
        double sz = (double)clustsizes[C[k] - 1];
        double pairs = sz * (sz - 1) / 2;                               /* binomial(sz,2) */
        double a = al + d1 * pairs;
        double b = be + BD[k * K + k] / 2;
        L1 += (d1 - 1) * BL[k * K + k] / 2 - pairs * lg_d1 + abratio + lgamma(a) - a * log(b);
    }
    double L2 = 0;
    for (int64_t k = 0; k < K; ++k)                                     /* mcmc.jl:39-53 */
        for (int64_t t = k + 1; t < K; ++t) {
            double pairs = (double)clustsizes[C[k] - 1] * (double)clustsizes[C[t] - 1];
            double z = ze + d2 * pairs;
            double g = ga + BD[k * K + t];
            L2 += (d2 - 1) * BL[k * K + t] - pairs * lg_d2 + zgratio + lgamma(z) - z * log(g);
        }
    free(C); free(slot); free(BD); free(BL);
    return L1 + (P->repulsion ? L2 : 0.0);                              /* mcmc.jl:54 */
}

/* loglik in stable arithmetic from exact fixed-point block sums (long double scalar part). */
double orc_loglik_stable(int64_t n, const int64_t *Dq, const int64_t *Lq, int eD, int eL,
                         const int64_t *clusts, const int64_t *clustsizes, const orc_params *P)
{
    int64_t K = 0;
    int64_t *C = malloc((size_t)n * sizeof(int64_t)), *slot = malloc((size_t)n * sizeof(int64_t));
    for (int64_t k = 0; k < n; ++k) { slot[k] = -1; if (clustsizes[k] > 0) { slot[k] = K; C[K++] = k + 1; } }
    /* int64 block sums can exceed 2^63 for a whole block (n^2 terms): accumulate in __int128 */
    __int128 *BD = calloc((size_t)(K * K), sizeof(__int128)), *BL = calloc((size_t)(K * K), sizeof(__int128));
    for (int64_t i = 0; i < n; ++i) {
        int64_t a = slot[clusts[i] - 1];
        for (int64_t j = 0; j < n; ++j) {
            int64_t b = slot[clusts[j] - 1];
            BD[a * K + b] += Dq[i * n + j];
            BL[a * K + b] += Lq[i * n + j];
        }
    }
    long double d1 = P->delta1, d2 = P->delta2, al = P->alpha, be = P->beta, ze = P->zeta, ga = P->gamma;
    long double lga = lgammal(al), lgz = lgammal(ze), lgd1 = lgammal(d1), lgd2 = lgammal(d2);
    long double lb = logl(be), lg = logl(ga);
    long double scD = ldexpl(1.0L, -eD), scL = ldexpl(1.0L, -eL);
    long double L1 = 0, L2 = 0;
    for (int64_t k = 0; k < K; ++k) {
        long double sz = (long double)clustsizes[C[k] - 1];
        long double pairs = sz * (sz - 1) / 2;
        long double a = al + d1 * pairs;
        long double bd = (long double)BD[k * K + k] * scD / 2, bl = (long double)BL[k * K + k] * scL / 2;
        /* αβratio + lgΓ(a) − a log b = [lgΓ(a) − lgΓ(α)] − δ1·pairs·log β − a·log1p(bd/β) */
        L1 += (d1 - 1) * bl - pairs * lgd1 + (lgammal(a) - lga) - d1 * pairs * lb - a * log1pl(bd / be);
    }
    for (int64_t k = 0; k < K; ++k)
        for (int64_t t = k + 1; t < K; ++t) {
            long double pairs = (long double)clustsizes[C[k] - 1] * (long double)clustsizes[C[t] - 1];
            long double z = ze + d2 * pairs;
            long double bd = (long double)BD[k * K + t] * scD, bl = (long double)BL[k * K + t] * scL;
            L2 += (d2 - 1) * bl - pairs * lgd2 + (lgammal(z) - lgz) - d2 * pairs * lg - z * log1pl(bd / ga);
        }
    free(C); free(slot); free(BD); free(BL);
    return (double)(L1 + (P->repulsion ? L2 : 0.0L));
}

/* logprior (src/mcmc.jl:58-78).  logpdf(Gamma(η, 1/σ), r) with shape η, scale 1/σ;
 * logpdf(Beta(u,v), p). */
double orc_logprior(int64_t n, const int64_t *clustsizes, double r, double p, const orc_params *P)
{
    int64_t K = 0;
    for (int64_t k = 0; k < n; ++k) K += clustsizes[k] > 0;
    double eta = P->eta, sigma = P->sigma, u = P->u, v = P->v;
    double lgam = eta * log(sigma) - lgamma(eta) + (eta - 1) * log(r) - sigma * r;
    double lbeta = lgamma(u + v) - lgamma(u) - lgamma(v) + (u - 1) * log(p) + (v - 1) * log(1 - p);
    double L = lgamma((double)K + 1) + (double)(n - K) * log(p) + (r * (double)K) * log(1 - p) -
               (double)K * lgamma(r) + lgam + lbeta;                    /* mcmc.jl:73 */
    for (int64_t k = 0; k < n; ++k)
        if (clustsizes[k] > 0) {
            double nj = (double)clustsizes[k];
            L += log(nj) + lgamma(nj + r - 1);                          /* mcmc.jl:75 */
        }
    return L;
}

/* sortlabels (src/utils.jl:69-74): relabel by order of first appearance (StatsBase.levelsmap). */
void orc_sortlabels(int64_t n, const int64_t *x, int64_t *y)
{
    int64_t *map = calloc((size_t)(n + 1), sizeof(int64_t));
    int64_t next = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (map[x[i]] == 0) map[x[i]] = ++next;
        y[i] = map[x[i]];
    }
    free(map);
}

/* counts += adjacencymatrix(clusts) (src/utils.jl:59-63, summed at src/mcmc.jl:560). */
void orc_cocluster_add(int64_t n, const int64_t *clusts, uint32_t *counts)
{
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j) counts[i * n + j] += (clusts[i] == clusts[j]);
}

/* matsum / vecsum (src/utils.jl:9-38) — plain sequential sums, for the restated value tests. */
double orc_matsum_idx(int64_t n, const double *x, const int64_t *inds1, int64_t n1, const int64_t *inds2,
                      int64_t n2)
{
    double ans = 0;
    for (int64_t i = 0; i < n1; ++i)
        for (int64_t j = 0; j < n2; ++j) ans += x[(inds1[i] - 1) + n * (inds2[j] - 1)]; /* column-major */
    return ans;
}
double orc_vecsum_idx(const double *x, const int64_t *inds, int64_t m)
{
    double ans = 0;
    for (int64_t i = 0; i < m; ++i) ans += x[inds[i] - 1];
    return ans;
}
