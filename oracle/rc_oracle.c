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
 * sample_logweights (src/utils.jl:4) draws for point i in sweep t are u(seed, t, i, key), one per
 * candidate, key = the candidate's cluster LABEL (1..n) for an existing cluster and 0 for the
 * new-cluster candidate, from Philox4x32-10 (Salmon et al., SC'11) keyed by the seed with the counter
 * (key, i, t_lo, t_hi).  Strictly inside (0,1).  (Keying by label, not by position in the candidate
 * list: m independent uniforms either way, but a candidate keeps its uniform when some other cluster
 * is born or dies earlier in the sweep, which is what lets the device resolver validate such changes
 * in batches.)
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
static int64_t draw_literal(double *lp, int64_t m, uint64_t seed, uint64_t sweep, uint64_t i,
                            const int64_t *cand, int64_t K_i)
{
    double mn = lp[0];
    for (int64_t k = 1; k < m; ++k) if (lp[k] < mn) mn = lp[k];
    int64_t best = 0;
    double bestv = 0;
    for (int64_t k = 0; k < m; ++k) {
        lp[k] -= mn;
        double u = orc_uniform(seed, sweep, i, k < K_i ? (uint64_t)cand[k] : 0u);  /* label; 0 = new cluster */
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
        int64_t K_i = 0;
        for (int64_t t = 0; t < n; ++t) K_i += clustsizes[t] > 0;
        int64_t k = draw_literal(w.logprobs, m, seed, sweep, (uint64_t)i, w.cand, K_i);  /* mcmc.jl:249 */
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
            double u = orc_uniform(seed, sweep, (uint64_t)i, (uint64_t)C_i[t]);
            v = v + (-log(-log(u)));
            if (t == 0 || v > bestv) { bestv = v; best = t; }
        }
        int64_t newlab = 0;
        if ((P->maxK == 0 || K_i < P->maxK) && K_i < n) {
            int64_t e = 0;
            while (clustsizes[e] != 0) ++e;
            newlab = e + 1;
            double v = log((double)(K_i + 1)) + r * log1mp;
            double u = orc_uniform(seed, sweep, (uint64_t)i, 0u);
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

/* The same sweep driven by a row-sum TABLE instead of the n×n matrices, for sizes at which the checker cannot hold
 * the matrices (BASELINE config 5: 2 × 8 GiB as int64).  Same loop, same arithmetic (stable_score, orc_uniform,
 * first-index argmax, smallest empty label) as orc_sweep_stable; only the source of matsum(·,[i],clust_k)
 * (src/mcmc.jl:210-213) differs:
 *   T_D / T_L  [nlab][n]  T[t][i] = Σ_j X[i,j]·[c_j = row_label[t]] under the labels at entry, j = i INCLUDED (D[i,i] as
 *                         stored — diag[i]; logD's diagonal is 0, types.jl:155); row_label ascending = the non-empty labels.
 *   A label change of point x (a → b) subtracts row x of each matrix from table row a and adds it to table row b.  The
 *   caller supplies the matrix rows of the points that are allowed to change: xs (ascending point indices, 0-based),
 *   XD / XL [nx][n] fixed point.  A change of any other point returns -3 (teacher forcing: the caller passes the set of
 *   points the implementation under test changed, so -3 is itself a parity failure).
 * Returns 0, or -2 (allocation), -3 (see above). */
int orc_sweep_table(int64_t n, int64_t nlab, const int64_t *row_label, const int64_t *T_D, const int64_t *T_L,
                    const int64_t *diag, int eD, int eL, const double *A, int64_t *clusts, int64_t *clustsizes,
                    int64_t *K, const orc_params *P, double r, double p, uint64_t seed, uint64_t sweep, int64_t nx,
                    const int64_t *xs, const int64_t *XD, const int64_t *XL, int64_t *n_changes)
{
    const int64_t cap = nlab + nx + 1;
    int64_t *TD = malloc((size_t)cap * (size_t)n * sizeof(int64_t)), *TL = malloc((size_t)cap * (size_t)n * sizeof(int64_t));
    int64_t *row_of = malloc((size_t)n * sizeof(int64_t));      /* label-1 -> table row, -1 = none */
    int64_t *act = malloc((size_t)(n + 1) * sizeof(int64_t));   /* non-empty labels, ascending (C_i of mcmc.jl:195) */
    int64_t *free_rows = malloc((size_t)cap * sizeof(int64_t));
    if (!TD || !TL || !row_of || !act || !free_rows) { free(TD); free(TL); free(row_of); free(act); free(free_rows); return -2; }
    memcpy(TD, T_D, (size_t)nlab * (size_t)n * sizeof(int64_t));
    memcpy(TL, T_L, (size_t)nlab * (size_t)n * sizeof(int64_t));
    memset(TD + nlab * n, 0, (size_t)(cap - nlab) * (size_t)n * sizeof(int64_t));
    memset(TL + nlab * n, 0, (size_t)(cap - nlab) * (size_t)n * sizeof(int64_t));
    for (int64_t k = 0; k < n; ++k) row_of[k] = -1;
    for (int64_t t = 0; t < nlab; ++t) row_of[row_label[t] - 1] = t;
    int64_t nfree = 0;
    for (int64_t t = cap - 1; t >= nlab; --t) free_rows[nfree++] = t;
    int64_t nact = 0;
    for (int64_t k = 0; k < n; ++k) if (clustsizes[k] > 0) act[nact++] = k + 1;
    const double scD = ldexp(1.0, -eD), scL = ldexp(1.0, -eL);
    const double logp = log(p), log1mp = log(1 - p);
    int64_t changes = 0;
    int rc = 0;
    for (int64_t i = 0; i < n && rc == 0; ++i) {
        const int64_t old = clusts[i];
        clustsizes[old - 1] -= 1;                                   /* mcmc.jl:193 */
        clusts[i] = -1;                                             /* mcmc.jl:194 */
        if (clustsizes[old - 1] == 0) {                             /* i was a singleton: its label leaves C_i */
            int64_t q = 0;
            while (act[q] != old) ++q;
            memmove(act + q, act + q + 1, (size_t)(nact - q - 1) * sizeof(int64_t));
            --nact;
        }
        const int64_t K_i = nact;
        int64_t best = 0;
        double bestv = 0;
        for (int64_t t = 0; t < K_i; ++t) {
            const int64_t c = act[t] - 1, row = row_of[c];
            int64_t sd = TD[row * n + i], sl = TL[row * n + i];
            if (c == old - 1) sd -= diag[i];                        /* i itself is not a member (clusts[i] = -1) */
            double v = stable_score(P, A, clustsizes[c], (double)sd * scD, (double)sl * scL, logp, r);
            double u = orc_uniform(seed, sweep, (uint64_t)i, (uint64_t)act[t]);
            v = v + (-log(-log(u)));
            if (t == 0 || v > bestv) { bestv = v; best = t; }
        }
        int64_t newlab = 0;
        if ((P->maxK == 0 || K_i < P->maxK) && K_i < n) {           /* mcmc.jl:198-203 */
            int64_t e = 0;
            while (clustsizes[e] != 0) ++e;
            newlab = e + 1;
            double v = log((double)(K_i + 1)) + r * log1mp;
            double u = orc_uniform(seed, sweep, (uint64_t)i, 0u);
            v = v + (-log(-log(u)));
            if (K_i == 0 || v > bestv) { bestv = v; best = K_i; }
        }
        const int64_t ci_new = (best < K_i) ? act[best] : newlab;
        if (clustsizes[ci_new - 1] == 0) {                          /* label (re)enters C_i, ascending order kept */
            int64_t q = 0;
            while (q < nact && act[q] < ci_new) ++q;
            memmove(act + q + 1, act + q, (size_t)(nact - q) * sizeof(int64_t));
            act[q] = ci_new;
            ++nact;
        }
        clusts[i] = ci_new;                                         /* mcmc.jl:250-252 */
        clustsizes[ci_new - 1] += 1;
        if (ci_new != old) {
            ++changes;
            int64_t lo = 0, hi = nx;
            while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (xs[mid] < i) lo = mid + 1; else hi = mid; }
            if (lo >= nx || xs[lo] != i) { rc = -3; break; }
            if (row_of[ci_new - 1] < 0) row_of[ci_new - 1] = free_rows[--nfree];   /* all-zero row */
            const int64_t ra = row_of[old - 1], rb = row_of[ci_new - 1];
            const int64_t *xd = XD + lo * n, *xl = XL + lo * n;
            for (int64_t j = 0; j < n; ++j) {
                TD[ra * n + j] -= xd[j]; TD[rb * n + j] += xd[j];
                TL[ra * n + j] -= xl[j]; TL[rb * n + j] += xl[j];
            }
            if (clustsizes[old - 1] == 0) { free_rows[nfree++] = ra; row_of[old - 1] = -1; }   /* exactly zero again */
        }
    }
    int64_t k = 0;
    for (int64_t t = 0; t < n; ++t) k += clustsizes[t] > 0;
    *K = k;
    if (n_changes) *n_changes = changes;
    free(TD); free(TL); free(row_of); free(act); free(free_rows);
    return rc;
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

/* ================================================================================================
 * Split–merge step: sample_labels! (src/mcmc.jl:356-479) and sample_labels_Gibbs_restricted!
 * (src/mcmc.jl:259-354), restated AS WRITTEN, including
 *   Q1  `state = finalstate` (mcmc.jl:470) rebinds the local name only: an accepted proposal, later proposals
 *       of the same iteration and the closing Gibbs sweep act on an object the caller never sees;
 *   Q2  `logprobs .+= minimum(logprobs)` (mcmc.jl:348): a no-op after sample_logweights' in-place shift in
 *       free mode (exp overflows beyond a gap of ~709), doubles the magnitude in forced mode;
 *   Q3  `L2_i = L2_ik_prime[1] + L2_ik_prime[2]` (mcmc.jl:331): the first two non-empty clusters.
 * Uniform stream of the MH block: Philox4x32-10 keyed (seed_lo, seed_hi ^ 0x4D485F52) with counter
 * (draw, mh_counter, iter_lo, iter_hi); draw 0/1 = chaperones (i = floor(u0 n); j = floor(u1 (n-1)), +1 if >= i),
 * 2 = acceptance, 4+q = launch assignment of the q-th element of S, 4+|S|+2|S|s+2q+k = k-th uniform of item q
 * in restricted scan s.
 * ============================================================================================== */
double orc_uniform_mh(uint64_t seed, uint64_t iter, uint64_t mh, uint64_t draw)
{
    uint32_t c[4] = {(uint32_t)draw, (uint32_t)mh, (uint32_t)iter, (uint32_t)(iter >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x4D485F52u);
    uint64_t bits = (((uint64_t)c[0] << 32) | c[1]) >> 12;
    return ((double)bits + 0.5) * 0x1p-52;
}

/* sum over the members (ascending index) of cluster `lab` of row x of M */
static double row_member_sum(int64_t n, const double *M, const int64_t *clusts, int64_t x, int64_t lab)
{
    double s = 0;
    for (int64_t y = 0; y < n; ++y)
        if (clusts[y] == lab) s += M[x * n + y];
    return s;
}

/* sample_labels_Gibbs_restricted! (mcmc.jl:259-354).  items: 0-based indices; cand: two 1-based labels;
 * final_clusts: NULL (free allocation) or the forced labels.  Returns log_transition_prob. */
static double restricted_scan(int64_t n, const double *D, const double *logD, int64_t *clusts, int64_t *clustsizes,
                              const orc_params *P, double r, double p, const int64_t *items, int64_t m,
                              const int64_t cand[2], const int64_t *final_clusts, uint64_t seed, uint64_t iter,
                              uint64_t mh, int64_t scan)
{
    const double d1 = P->delta1, d2 = P->delta2, al = P->alpha, be = P->beta, ze = P->zeta, ga = P->gamma;
    const double abratio = al * log(be) - lgamma(al), zgratio = ze * log(ga) - lgamma(ze);   /* :293-294 */
    const double lg_d1 = lgamma(d1), lg_d2 = lgamma(d2), logp = log(p);                     /* :295-297 */
    const double rep = P->repulsion ? 1.0 : 0.0;
    /* C = findall(clustsizes .> 0), frozen at entry (:273); only C[1], C[2] and the candidates are ever used */
    int64_t C1 = 0, C2 = 0;
    for (int64_t k = 0; k < n && C2 == 0; ++k)
        if (clustsizes[k] > 0) { if (C1 == 0) C1 = k + 1; else C2 = k + 1; }
    double ltp = 0;                                                                        /* :285 */
    for (int64_t q = 0; q < m; ++q) {
        const int64_t x = items[q];
        clustsizes[clusts[x] - 1] -= 1;                                                    /* :303 */
        clusts[x] = -1;                                                                    /* :304 */
        double L1[2], lpr[2], L2[2], logprobs[2], L2p_c[2];
        for (int k = 0; k < 2; ++k) {                                                      /* :307-312, :321-326 */
            const int64_t lab = cand[k];
            const double sz = (double)clustsizes[lab - 1];
            const double sD = row_member_sum(n, D, clusts, x, lab);
            const double sL = row_member_sum(n, logD, clusts, x, lab);
            const double a_i = al + d1 * sz, b_i = be + sD;
            const double z_i = ze + d2 * sz, g_i = ga + sD;
            L1[k] = lgamma(a_i) + abratio - a_i * log(b_i) + (d1 - 1) * sL - sz * lg_d1;
            lpr[k] = log(sz + 1) + logp + log(sz - 1 + r) - log(sz);
            L2p_c[k] = lgamma(z_i) - z_i * log(g_i) + zgratio + (d2 - 1) * sL - sz * lg_d2;  /* :328-329 for this label */
        }
        double L2p_first[2];
        const int64_t firsts[2] = {C1, C2};
        for (int t = 0; t < 2; ++t) {                                                      /* :313-319, :327-331 */
            const int64_t lab = firsts[t];
            if (lab == 0) { L2p_first[t] = 0; continue; }  /* fewer than two clusters: BoundsError in Julia; unreachable */
            if (lab == cand[0]) { L2p_first[t] = L2p_c[0]; continue; }
            if (lab == cand[1]) { L2p_first[t] = L2p_c[1]; continue; }
            const double sz = (double)clustsizes[lab - 1];
            const double sD = row_member_sum(n, D, clusts, x, lab);
            const double sL = row_member_sum(n, logD, clusts, x, lab);
            const double z_i = ze + d2 * sz, g_i = ga + sD;
            L2p_first[t] = lgamma(z_i) - z_i * log(g_i) + zgratio + (d2 - 1) * sL - sz * lg_d2;
        }
        const double L2_i = L2p_first[0] + L2p_first[1];                                   /* :331 (Q3) */
        for (int k = 0; k < 2; ++k) {
            L2[k] = L2_i - L2p_c[k];                                                       /* :333 */
            logprobs[k] = lpr[k] + (L1[k] + (rep != 0.0 ? L2[k] : 0.0));                   /* :335 */
        }
        int k;
        if (!final_clusts) {                                                               /* :336-338 */
            /* sample_logweights mutates its argument (utils.jl:3) */
            const double mn = logprobs[0] < logprobs[1] ? logprobs[0] : logprobs[1];
            logprobs[0] -= mn; logprobs[1] -= mn;
            const uint64_t base = 4 + (uint64_t)m + 2 * (uint64_t)m * (uint64_t)scan + 2 * (uint64_t)q;
            const double g0 = -log(-log(orc_uniform_mh(seed, iter, mh, base))) + logprobs[0];
            const double g1 = -log(-log(orc_uniform_mh(seed, iter, mh, base + 1))) + logprobs[1];
            k = (g1 > g0) ? 1 : 0;                                                         /* argmax, first wins ties */
        } else {                                                                           /* :339-342 */
            k = (final_clusts[x] == cand[0]) ? 0 : 1;
        }
        clusts[x] = cand[k];                                                               /* :344 */
        clustsizes[cand[k] - 1] += 1;                                                      /* :345 */
        const double mn = logprobs[0] < logprobs[1] ? logprobs[0] : logprobs[1];          /* :348 (Q2) */
        /* Julia's minimum propagates NaN */
        const double mnn = (logprobs[0] != logprobs[0] || logprobs[1] != logprobs[1]) ? NAN : mn;
        double pr[2] = {exp(logprobs[0] + mnn), exp(logprobs[1] + mnn)};                   /* :349 */
        const double tot = pr[0] + pr[1];
        ltp += log(pr[k] / tot);                                                           /* :350-351 */
    }
    return ltp;
}

typedef struct {
    int32_t accept, split, skipped, pad;
    int64_t i, j, nS;
    double log_prior_ratio, log_lik_ratio, log_proposal_ratio, log_u;
} orc_mh_info;

/* One proposal of the MH loop body (mcmc.jl:374-473) on (clusts, clustsizes, K).  On acceptance the arrays are
 * REPLACED by the final state (the caller decides what that means: Q1).  loglik_mode 0 = literal, 1 = stable
 * (needs Dq/Lq; pass NULL otherwise). */
int orc_mh_proposal(int64_t n, const double *D, const double *logD, const int64_t *Dq, const int64_t *Lq, int eD, int eL,
                    int64_t *clusts, int64_t *clustsizes, int64_t *K, const orc_params *P, double r, double p,
                    int64_t numGibbs, uint64_t seed, uint64_t iter, uint64_t mh, int loglik_mode, orc_mh_info *info)
{
    memset(info, 0, sizeof(*info));
    /* i, j = sample(1:n, 2, replace=false)  (:379) */
    int64_t i = (int64_t)floor(orc_uniform_mh(seed, iter, mh, 0) * (double)n);
    int64_t j = (int64_t)floor(orc_uniform_mh(seed, iter, mh, 1) * (double)(n - 1));
    if (i >= n) i = n - 1;
    if (j >= n - 1) j = n - 2;
    if (j >= i) j += 1;
    info->i = i; info->j = j;
    const int64_t ci = clusts[i], cj = clusts[j];
    int64_t nonempty = 0;
    for (int64_t k = 0; k < n; ++k) nonempty += clustsizes[k] > 0;
    if (P->maxK > 0 && ci == cj && nonempty >= P->maxK) { info->skipped = 1; return 0; }   /* :384-386 */
    int64_t *S = malloc((size_t)n * sizeof(int64_t));
    int64_t m = 0;
    for (int64_t k = 0; k < n; ++k)
        if ((clusts[k] == ci || clusts[k] == cj) && k != i && k != j) S[m++] = k;          /* :389-390 */
    info->nS = m;
    int64_t *claunch = malloc((size_t)n * sizeof(int64_t)), *szlaunch = malloc((size_t)n * sizeof(int64_t));
    memcpy(claunch, clusts, (size_t)n * sizeof(int64_t));
    memcpy(szlaunch, clustsizes, (size_t)n * sizeof(int64_t));
    int64_t Klaunch = *K;
    if (ci == cj) {                                                                         /* :396-401 */
        int64_t e = 0;
        while (clustsizes[e] != 0) ++e;
        claunch[i] = e + 1;
        szlaunch[ci - 1] -= 1;
        szlaunch[e] += 1;
        Klaunch = *K + 1;
    }
    const int64_t cand[2] = {claunch[i], claunch[j]};                                       /* :402 */
    for (int64_t q = 0; q < m; ++q) {                                                       /* :403-407 */
        const int64_t k = S[q];
        claunch[k] = cand[orc_uniform_mh(seed, iter, mh, 4 + (uint64_t)q) < 0.5 ? 0 : 1];
        szlaunch[clusts[k] - 1] -= 1;
        szlaunch[claunch[k] - 1] += 1;
    }
    for (int64_t s = 0; s < numGibbs; ++s)                                                  /* :411-414 */
        restricted_scan(n, D, logD, claunch, szlaunch, P, r, p, S, m, cand, NULL, seed, iter, mh, s);
    int64_t *cfinal, *szfinal, Kfinal;
    double log_prior_ratio, log_proposal_ratio;
    if (ci == cj) {                                                                         /* split :416-434 */
        info->split = 1;
        const double ltp = restricted_scan(n, D, logD, claunch, szlaunch, P, r, p, S, m, cand, NULL, seed, iter, mh, numGibbs);
        cfinal = claunch; szfinal = szlaunch; Kfinal = Klaunch;
        log_prior_ratio = log((double)(*K + 1)) + r * log(1 - p) - log(p) - lgamma(r) +
                          lgamma((double)szfinal[cfinal[i] - 1] - 1 + r) + lgamma((double)szfinal[cfinal[j] - 1] - 1 + r) +
                          log((double)szfinal[cfinal[i] - 1]) + log((double)szfinal[cfinal[j] - 1]) +
                          -(lgamma((double)clustsizes[ci - 1] - 1 + r) + log((double)clustsizes[ci - 1]));
        log_proposal_ratio = ltp;
    } else {                                                                                /* merge :435-459 */
        cfinal = malloc((size_t)n * sizeof(int64_t)); szfinal = malloc((size_t)n * sizeof(int64_t));
        memcpy(cfinal, claunch, (size_t)n * sizeof(int64_t));
        memcpy(szfinal, szlaunch, (size_t)n * sizeof(int64_t));
        Kfinal = Klaunch;
        int64_t sz_clust_i = 0;
        for (int64_t k = 0; k < n; ++k)
            if (cfinal[k] == ci) { cfinal[k] = cj; ++sz_clust_i; }
        szfinal[ci - 1] = 0;
        szfinal[cj - 1] += sz_clust_i;
        Kfinal -= 1;
        log_prior_ratio = -(log((double)*K) + r * log(1 - p) - log(p) - lgamma(r)) +
                          lgamma((double)szfinal[cj - 1] - 1 + r) + log((double)szfinal[cj - 1]) +
                          -(lgamma((double)clustsizes[ci - 1] - 1 + r) + lgamma((double)clustsizes[cj - 1] - 1 + r) +
                            log((double)clustsizes[ci - 1]) + log((double)clustsizes[cj - 1]));
        const double ltp = restricted_scan(n, D, logD, claunch, szlaunch, P, r, p, S, m, cand, clusts, seed, iter, mh, numGibbs);
        log_proposal_ratio = -ltp;
    }
    double llr;                                                                             /* :462-464 */
    if (loglik_mode == 0)
        llr = orc_loglik_literal(n, D, logD, cfinal, szfinal, P) - orc_loglik_literal(n, D, logD, clusts, clustsizes, P);
    else
        llr = orc_loglik_stable(n, Dq, Lq, eD, eL, cfinal, szfinal, P) - orc_loglik_stable(n, Dq, Lq, eD, eL, clusts, clustsizes, P);
    const double x = log_prior_ratio + llr - log_proposal_ratio;
    /* minimum([0, x]) propagates NaN (:467-468); log(u) < NaN is false */
    const double lar = (x != x) ? NAN : (x < 0 ? x : 0.0);
    const double lu = log(orc_uniform_mh(seed, iter, mh, 2));
    info->log_prior_ratio = log_prior_ratio; info->log_lik_ratio = llr; info->log_proposal_ratio = log_proposal_ratio;
    info->log_u = lu;
    if (lu < lar) {                                                                         /* :469-472 */
        memcpy(clusts, cfinal, (size_t)n * sizeof(int64_t));
        memcpy(clustsizes, szfinal, (size_t)n * sizeof(int64_t));
        *K = Kfinal;
        info->accept = 1;
    }
    if (cfinal != claunch) { free(cfinal); free(szfinal); }
    free(claunch); free(szlaunch); free(S);
    return 0;
}

/* sample_labels!(data, state, params, options) AS WRITTEN (mcmc.jl:356-479): numMH proposals, then the full
 * Gibbs sweep.  Because of Q1 the caller's (clusts, clustsizes, K) are mutated by the sweep only when NO proposal
 * was accepted; after an acceptance everything (later proposals, the sweep) happens on a private object.
 * sweep_mode: 0 literal, 1 stable (fixed-point).  accept/split: numMH flags each (undefined entries stay 0, as the
 * reference's `fill(false, numMH)`).  Returns the number of accepted proposals. */
int orc_sample_labels(int64_t n, const double *D, const double *logD, const int64_t *Dq, const int64_t *Lq, int eD, int eL,
                      const double *A, int64_t *clusts, int64_t *clustsizes, int64_t *K, const orc_params *P, double r,
                      double p, int64_t numMH, int64_t numGibbs, uint64_t seed, uint64_t iter, int mode, uint8_t *accept,
                      uint8_t *split)
{
    int64_t *c = clusts, *s = clustsizes, Kloc = *K;
    int64_t *pc = NULL, *ps = NULL;
    int naccept = 0;
    for (int64_t mh = 0; mh < numMH; ++mh) {
        accept[mh] = 0; split[mh] = 0;
        if (!pc) {  /* still the caller's object: propose on a scratch copy so that an acceptance can rebind */
            pc = malloc((size_t)n * sizeof(int64_t)); ps = malloc((size_t)n * sizeof(int64_t));
            memcpy(pc, c, (size_t)n * sizeof(int64_t)); memcpy(ps, s, (size_t)n * sizeof(int64_t));
        }
        orc_mh_info info;
        int64_t Ktmp = Kloc;
        int64_t *wc = (c == clusts) ? pc : c, *ws = (s == clustsizes) ? ps : s;
        if (c == clusts) { memcpy(pc, c, (size_t)n * sizeof(int64_t)); memcpy(ps, s, (size_t)n * sizeof(int64_t)); }
        orc_mh_proposal(n, D, logD, Dq, Lq, eD, eL, wc, ws, &Ktmp, P, r, p, numGibbs, seed, iter, (uint64_t)mh, mode, &info);
        split[mh] = (uint8_t)info.split;
        if (info.accept) {
            accept[mh] = 1;
            ++naccept;
            c = wc; s = ws; Kloc = Ktmp;   /* `state = finalstate`: the local name now refers to the private object */
        }
    }
    /* final Gibbs scan on whatever `state` names now (:477) */
    int64_t Ksw = Kloc;
    if (mode == 0) orc_sweep_literal(n, D, logD, c, s, &Ksw, P, r, p, seed, iter, 0);
    else orc_sweep_stable(n, Dq, Lq, eD, eL, A, c, s, &Ksw, P, r, p, seed, iter, NULL);
    if (c == clusts) *K = Ksw;  /* the caller's object was swept; otherwise it is untouched (Q1) */
    free(pc); free(ps);
    return naccept;
}


/* MCMCData(points) (src/types.jl:159-162): D = pairwise(Euclidean(), makematrix(pnts), dims=2).  Distances.jl
 * (third party, not under the reference checkout; restated from its published algorithm for a single matrix):
 * sa2[i] = Σ_k a[k,i]², R = a'a, r[i,j] = sqrt(max(sa2[i] + sa2[j] − 2R[i,j], 0)) for i > j, mirrored, diagonal 0.
 * pts: n×dim row-major (point i = pts[i*dim .. ]).  R's BLAS summation order is unspecified; plain ascending here. */
void orc_pairwise_euclidean(int64_t n, int64_t dim, const double *pts, double *D)
{
    double *sa2 = malloc((size_t)n * sizeof(double));
    for (int64_t i = 0; i < n; ++i) {
        double s = 0;
        for (int64_t k = 0; k < dim; ++k) s += pts[i * dim + k] * pts[i * dim + k];
        sa2[i] = s;
    }
    for (int64_t j = 0; j < n; ++j) {
        D[j * n + j] = 0;
        for (int64_t i = j + 1; i < n; ++i) {
            double r = 0;
            for (int64_t k = 0; k < dim; ++k) r += pts[i * dim + k] * pts[j * dim + k];
            double d2 = sa2[i] + sa2[j] - 2 * r;
            double d = sqrt(d2 > 0 ? d2 : 0);
            D[i * n + j] = d;
            D[j * n + i] = d;
        }
    }
    free(sa2);
}


/* ================================================================================================
 * Point estimation (src/pointestimate.jl) and clustering comparison (src/summaries.jl:12-23).
 *
 * The pairwise measures come from Clustering.jl (third party, Project.toml:22 pins 0.13.5 / 0.14 / 0.15; not under
 * the reference checkout).  Restated from their published definitions on the contingency table n_ij of the two
 * labelings, row sums a_i, column sums b_j, N points:
 *   randindex (Hubert & Arabie 1985):  t1 = C(N,2), t2 = Σ n_ij², t3 = (Σ a_i² + Σ b_j²)/2,
 *       nc = (N(N²+1) − (N+1)Σa_i² − (N+1)Σb_j² + 2 Σa_i² Σb_j² / N) / (2(N−1)),
 *       A = t1 + t2 − t3, D = t3 − t2;  ARI = (A − nc)/(t1 − nc) (0 if t1 == nc), RI = A/t1, Mirkin = D/t1,
 *       Hubert = (A − D)/t1
 *   mutualinfo(normed=false): I = Σ (n_ij/N) log(N n_ij / (a_i b_j));   normed: 2I/(H(a) + H(b))
 *   varinfo (Meilă 2007):     VI = H(a) + H(b) − 2I
 *   entropy(counts/N):        H(a) = −Σ (a_i/N) log(a_i/N)
 * Every quantity is evaluated from the four sums  Σ n_ij log n_ij, Σ a_i log a_i, Σ b_j log b_j, Σ n_ij²  so that
 * the GPU path (same sums, different summation order) agrees to rounding.
 * ============================================================================================== */

typedef struct {
    double ari, ri, mirkin, hubert; /* randindex(a, b) */
    double mi, nmi, vi;             /* mutualinfo(normed=false), mutualinfo(normed=true), varinfo */
    double ha, hb;                  /* entropy(counts(a)/N), entropy(counts(b)/N) */
    double id, nid;                 /* infodist(normalised=false / true), pointestimate.jl:89-99 */
} orc_pair_measures;

static int cmp_i64(const void *x, const void *y)
{
    int64_t u = *(const int64_t *)x, v = *(const int64_t *)y;
    return (u > v) - (u < v);
}

/* labels: any positive integers ≤ n (as in the reference: 1..n) */
void orc_pair_measures_eval(int64_t n, const int64_t *a, const int64_t *b, orc_pair_measures *out)
{
    int64_t *ca = calloc((size_t)n + 1, sizeof(int64_t)), *cb = calloc((size_t)n + 1, sizeof(int64_t));
    /* contingency table through a sort of the (a, b) key: no K×K storage */
    int64_t *key = malloc((size_t)n * sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) { ca[a[i]]++; cb[b[i]]++; key[i] = a[i] * (n + 1) + b[i]; }
    qsort(key, (size_t)n, sizeof(int64_t), cmp_i64);
    double e_ab = 0, e_a = 0, e_b = 0, t2 = 0, nis = 0, njs = 0;
    for (int64_t i = 0; i < n;) {
        int64_t j = i;
        while (j < n && key[j] == key[i]) ++j;
        double c = (double)(j - i);
        e_ab += c * log(c);
        t2 += c * c;
        i = j;
    }
    for (int64_t l = 1; l <= n; ++l) {
        if (ca[l]) { e_a += (double)ca[l] * log((double)ca[l]); nis += (double)ca[l] * (double)ca[l]; }
        if (cb[l]) { e_b += (double)cb[l] * log((double)cb[l]); njs += (double)cb[l] * (double)cb[l]; }
    }
    const double N = (double)n, logN = log(N);
    const double t1 = N * (N - 1) / 2, t3 = 0.5 * (nis + njs);
    const double nc = (N * (N * N + 1) - (N + 1) * nis - (N + 1) * njs + 2 * (nis * njs) / N) / (2 * (N - 1));
    const double A = t1 + t2 - t3, Dd = -t2 + t3;
    out->ari = (t1 == nc) ? 0.0 : (A - nc) / (t1 - nc);
    out->ri = A / t1;
    out->mirkin = Dd / t1;
    out->hubert = (A - Dd) / t1;
    out->ha = logN - e_a / N;
    out->hb = logN - e_b / N;
    out->mi = (e_ab - e_a - e_b) / N + logN;
    out->nmi = 2 * out->mi / (out->ha + out->hb);
    out->vi = out->ha + out->hb - 2 * out->mi;
    const double hmax = out->ha > out->hb ? out->ha : out->hb;
    out->id = hmax - out->mi;
    out->nid = 1 - out->mi / hmax;
    free(ca); free(cb); free(key);
}

/* loss kinds of getpointestimate(method="MPEL") (pointestimate.jl:38-47): 0 binder = randindex[3], 1 omARI = 1 − ARI,
 * 2 VI = varinfo, 3 ID = infodist(normalised=false) */
static double pick_loss(const orc_pair_measures *m, int kind)
{
    switch (kind) {
    case 0: return m->mirkin;
    case 1: return 1 - m->ari;
    case 2: return m->vi;
    default: return m->id;
    }
}

/* The MPEL search (pointestimate.jl:49-58): upper-triangle losses, symmetrised, column sums, first argmin.
 * samples: m×n row-major.  lossmatrix (m×m, may be NULL), colsum (m).  Returns the 0-based argmin. */
int64_t orc_mpel(int64_t m, int64_t n, const int64_t *samples, int kind, double *lossmatrix, double *colsum)
{
    double *L = lossmatrix ? lossmatrix : calloc((size_t)(m * m), sizeof(double));
    for (int64_t i = 0; i < m * m; ++i) L[i] = 0;
    orc_pair_measures pm;
    for (int64_t i = 0; i < m; ++i)
        for (int64_t j = i + 1; j < m; ++j) {
            orc_pair_measures_eval(n, samples + i * n, samples + j * n, &pm);
            L[i * m + j] = pick_loss(&pm, kind);
            L[j * m + i] = L[i * m + j];
        }
    int64_t best = 0;
    for (int64_t j = 0; j < m; ++j) {
        double s = 0;
        for (int64_t i = 0; i < m; ++i) s += L[i * m + j];
        colsum[j] = s;
        if (s < colsum[best]) best = j;
    }
    if (!lossmatrix) free(L);
    return best;
}


/* ================================================================================================
 * Scalar updates of the chain: sample_r (src/mcmc.jl:94-136) and sample_p (src/mcmc.jl:147-155).
 *
 * The reference draws through Distributions.jl (truncated Normal, Beta) on Julia's global RNG; neither the
 * stream nor Distributions' internal algorithms can be reproduced, so the build defines its own exact samplers
 * on its counter-based uniform stream (DESIGN.md "scalar stream"):
 *   uniform  u(iter, kind, d) = Philox4x32-10, key (seed_lo, seed_hi ^ 0x52505F5F), counter (d, kind, iter_lo,
 *            iter_hi), u = (top 52 bits + 0.5)·2^-52;  kind 0 = r update, 1 = p update;  d counts the draws
 *   normal   Box–Muller: z = sqrt(-2 log u_d) · cos(2π u_{d+1})  (two draws each)
 *   truncated(Normal(mu, sd), 0, Inf): redraw until mu + sd·z >= 0
 *   Gamma(a, 1), a >= 1: Marsaglia & Tsang (2000): d = a − 1/3, c = 1/sqrt(9d); z normal, v = (1 + cz)³,
 *            accept if v > 0 and log u < z²/2 + d − dv + d log v;  a < 1: Gamma(a+1)·u^(1/a)
 *   Beta(a, b) = X/(X+Y), X ~ Gamma(a), Y ~ Gamma(b)
 * The log-densities are the reference's expressions literally.
 * ============================================================================================== */
typedef struct { uint64_t seed, iter; uint32_t kind; uint64_t draw; } orc_scalar_stream;

static double ss_uniform(orc_scalar_stream *s)
{
    uint32_t c[4] = {(uint32_t)s->draw, s->kind, (uint32_t)s->iter, (uint32_t)(s->iter >> 32)};
    s->draw++;
    philox4x32_10(c, (uint32_t)s->seed, (uint32_t)(s->seed >> 32) ^ 0x52505F5Fu);
    uint64_t bits = (((uint64_t)c[0] << 32) | c[1]) >> 12;
    return ((double)bits + 0.5) * 0x1p-52;
}

static double ss_normal(orc_scalar_stream *s)
{
    const double u1 = ss_uniform(s), u2 = ss_uniform(s);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

static double ss_gamma(orc_scalar_stream *s, double a)
{
    double boost = 1.0;
    if (a < 1.0) { boost = pow(ss_uniform(s), 1.0 / a); a += 1.0; }
    const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        const double z = ss_normal(s);
        const double t = 1.0 + c * z;
        const double u = ss_uniform(s);
        if (t <= 0) continue;
        const double v = t * t * t;
        if (log(u) < 0.5 * z * z + d - d * v + d * log(v)) return d * v * boost;
    }
}

double orc_scalar_uniform(uint64_t seed, uint64_t iter, uint32_t kind, uint64_t draw)
{
    orc_scalar_stream s = {seed, iter, kind, draw};
    return ss_uniform(&s);
}

/* logpdf(truncated(Normal(mu, sd), lower = 0, upper = Inf), x), x >= 0 */
static double logpdf_truncnorm0(double x, double mu, double sd)
{
    const double z = (x - mu) / sd;
    return -0.5 * z * z - log(sd) - 0.91893853320467274178 - log(0.5 * erfc(-(mu / sd) * 0.70710678118654752440));
}

/* sample_r (mcmc.jl:94-136).  C: sizes of the K non-empty clusters (ascending label order).  Returns the new r,
 * *accept = 0/1. */
double orc_sample_r(uint64_t seed, uint64_t iter, double r, double p, const int64_t *C, int64_t K, double eta,
                    double sigma, double proposalsd_r, int *accept)
{
    orc_scalar_stream s = {seed, iter, 0, 0};
    double rc;
    do rc = r + proposalsd_r * ss_normal(&s); while (rc < 0);                                        /* :105-110 */
    double lpc = (eta - 1) * log(rc) + (double)K * (rc * log(1 - p) - lgamma(rc)) - rc * sigma;      /* :118 */
    double lpo = (eta - 1) * log(r) + (double)K * (r * log(1 - p) - lgamma(r)) - r * sigma;          /* :119 */
    for (int64_t k = 0; k < K; ++k) {                                                                /* :120-123 */
        lpc = lpc + lgamma((double)(C[k] - 1) + rc);
        lpo = lpo + lgamma((double)(C[k] - 1) + r);
    }
    const double lpr = logpdf_truncnorm0(rc, r, proposalsd_r) - logpdf_truncnorm0(r, rc, proposalsd_r); /* :125-126 */
    double bound = lpc - lpo - lpr;
    if (bound > 0) bound = 0;                                                                        /* minimum([0, …]) */
    *accept = log(ss_uniform(&s)) < bound;                                                           /* :131-135 */
    return *accept ? rc : r;
}

/* sample_p (mcmc.jl:147-155): rand(Beta(n − K + u, r K + v)) */
double orc_sample_p(uint64_t seed, uint64_t iter, int64_t K, int64_t n, double r, double u, double v)
{
    orc_scalar_stream s = {seed, iter, 1, 0};
    const double x = ss_gamma(&s, (double)(n - K) + u);
    const double y = ss_gamma(&s, r * (double)K + v);
    return x / (x + y);
}
