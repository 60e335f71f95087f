// rc_run_chain — the iteration loop of runsampler (/root/reference/src/mcmc.jl:533-556) as native host code around
// the device sweep: sample_r!, sample_p!, sample_labels! (split–merge proposals + Gibbs sweep), recording rule.
// Included at the end of redclust_hip.hip (same translation unit).
//
// Why native: the loop is scalar work between kernels of ~100 µs; run from an interpreter it costs more than the
// sweep it drives.  Here one iteration of the host side is: wait for sweep t (its K and cluster sizes arrive in
// host-mapped memory), two scalar draws, launch sweep t+1 — while the row reduction of sweep t+1 has been running on
// the other stream since sweep t was launched.
//
// Scalar stream: Philox4x32-10 keyed (seed_lo, seed_hi ^ 0x52505F5F), counter (draw, kind, iter_lo,
// iter_hi); kind 0 = r update, 1 = p update.  Normal: Box–Muller; truncated Normal: redraw; Gamma: Marsaglia–Tsang;
// Beta = Gamma ratio.  (The reference samples through Distributions.jl on Julia's global RNG — same distributions,
// irreproducible stream.)

namespace chain {

struct Stream {
    uint64_t seed, iter;
    uint32_t kind;
    uint64_t draw;
    double uniform()
    {
        uint32_t c[4] = {(uint32_t)draw, kind, (uint32_t)iter, (uint32_t)(iter >> 32)};
        ++draw;
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ 0x52505F5Fu;
        for (int r = 0; r < 10; ++r) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
            c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        const uint64_t bits = (((uint64_t)c[0] << 32) | c[1]) >> 12;
        return ((double)bits + 0.5) * 0x1p-52;
    }
    double normal()
    {
        const double u1 = uniform(), u2 = uniform();
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925286766559 * u2);
    }
    double gamma(double a)
    {
        double boost = 1.0;
        if (a < 1.0) { boost = std::pow(uniform(), 1.0 / a); a += 1.0; }
        const double d = a - 1.0 / 3.0, c = 1.0 / std::sqrt(9.0 * d);
        for (;;) {
            const double z = normal();
            const double t = 1.0 + c * z;
            const double u = uniform();
            if (t <= 0) continue;
            const double v = t * t * t;
            if (std::log(u) < 0.5 * z * z + d - d * v + d * std::log(v)) return d * v * boost;
        }
    }
};

// logpdf(truncated(Normal(mu, sd), lower = 0, upper = Inf), x)
static double logpdf_truncnorm0(double x, double mu, double sd)
{
    const double z = (x - mu) / sd;
    return -0.5 * z * z - std::log(sd) - 0.91893853320467274178 - std::log(0.5 * std::erfc(-(mu / sd) * 0.70710678118654752440));
}

// sample_r (mcmc.jl:94-136); C = sizes of the non-empty clusters in ascending label order
static double sample_r(uint64_t seed, uint64_t iter, double r, double p, const std::vector<int64_t> &C, double eta,
                       double sigma, double proposalsd_r, bool *accept)
{
    Stream s{seed, iter, 0, 0};
    const double K = (double)C.size();
    double rc;
    do rc = r + proposalsd_r * s.normal(); while (rc < 0);
    double lpc = (eta - 1) * std::log(rc) + K * (rc * std::log(1 - p) - std::lgamma(rc)) - rc * sigma;
    double lpo = (eta - 1) * std::log(r) + K * (r * std::log(1 - p) - std::lgamma(r)) - r * sigma;
    for (int64_t nk : C) {
        lpc = lpc + std::lgamma((double)(nk - 1) + rc);
        lpo = lpo + std::lgamma((double)(nk - 1) + r);
    }
    const double lpr = logpdf_truncnorm0(rc, r, proposalsd_r) - logpdf_truncnorm0(r, rc, proposalsd_r);
    double bound = lpc - lpo - lpr;
    if (bound > 0) bound = 0;
    *accept = std::log(s.uniform()) < bound;
    return *accept ? rc : r;
}

// sample_p (mcmc.jl:147-155)
static double sample_p(uint64_t seed, uint64_t iter, int64_t K, int64_t n, double r, double u, double v)
{
    Stream s{seed, iter, 1, 0};
    const double x = s.gamma((double)(n - K) + u);
    const double y = s.gamma(r * (double)K + v);
    return x / (x + y);
}

// cluster sizes in ascending label order from the host-mapped sweep summary (valid after sync_and_check)
static void sizes_by_label(rc_ctx *c, std::vector<int64_t> &C)
{
    std::vector<std::pair<int, int>> bl;
    for (int k = 0; k < c->kcap; ++k)
        if (c->hsum->size_label[2 * k] > 0 && c->hsum->size_label[2 * k + 1] > 0)
            bl.push_back({c->hsum->size_label[2 * k + 1], c->hsum->size_label[2 * k]});
    std::sort(bl.begin(), bl.end());
    C.clear();
    for (auto &e : bl) C.push_back(e.second);
}

// ---- asynchronous recorder -------------------------------------------------------------------------------------
// A recorded sample (mcmc.jl:546-553) needs the labels, the block sums for loglik and the cluster sizes of the state
// an iteration ends in.  Instead of stalling the loop on them, the device part (label snapshot into the co-clustering
// queue + copy to pinned memory, k_blocksums + copy) is enqueued on stream A in front of the next sweep, and the host
// part (sortlabels, the scalar part of loglik, logprior) runs after that sweep has been launched, i.e. under it.
struct Pending {
    bool active = false;
    int64_t j = 0;
    double r = 0, p = 0;
    int hi = 0, K = 0;
    std::vector<int> ssize, slabel;
};
constexpr int REC_SLOT = RC_REC_SLOTS - 1;  // slot 0 belongs to rc_loglik (split–merge proposals call it)

// the state to record must be complete and its summary visible to the host (sync_and_check done)
static int32_t record_enqueue(rc_ctx *c, Pending &R, bool want_labels)
{
    int32_t rc = ensure_counts(c);
    if (rc != RC_OK) return rc;
    R.hi = std::max(1, std::min(c->kcap, c->hsum->slot_hi));
    R.K = c->hsum->K;
    rc = ensure_pinned(c, R.hi);
    if (rc != RC_OK) return rc;
    R.ssize.resize((size_t)c->kcap); R.slabel.resize((size_t)c->kcap);
    for (int k = 0; k < c->kcap; ++k) {
        R.ssize[(size_t)k] = c->hsum->size_label[2 * k];
        R.slabel[(size_t)k] = c->hsum->size_label[2 * k + 1];
    }
    unsigned short *row = c->snap + (size_t)c->snap_cnt * c->ldc;
    rc = order_A_after_sweeps(c);
    if (rc != RC_OK) return rc;
    k_snapshot<<<(c->ldc + 255) / 256, 256, 0, c->sA>>>(c->slot_of, c->pi, c->n, c->ldc, row);
    HIPCHK(c, hipGetLastError());
    if (want_labels) HIPCHK(c, hipMemcpyAsync(c->pinLab[REC_SLOT], row, (size_t)c->n * sizeof(unsigned short), hipMemcpyDeviceToHost, c->sA));
    if (++c->snap_cnt == RC_CC_BATCH) {
        rc = flush_counts(c);
        if (rc != RC_OK) return rc;
    }
    rc = loglik_enqueue(c, R.hi, c->pinB[REC_SLOT]);
    if (rc != RC_OK) return rc;
    HIPCHK(c, hipEventRecord(c->pinEv[REC_SLOT], c->sA));
    return RC_OK;
}

static int32_t record_finish(rc_ctx *c, Pending &R, rc_chain_outputs *out)
{
    HIPCHK(c, hipEventSynchronize(c->pinEv[REC_SLOT]));
    const int64_t j = R.j;
    if (out->clusts) {
        // sortlabels (utils.jl:69-74): relabel by order of first appearance; the snapshot is in the caller's point order
        int64_t *dst = out->clusts + (size_t)j * c->n;
        std::vector<int> map((size_t)c->kcap, 0);
        int next = 0;
        const unsigned short *lab = c->pinLab[REC_SLOT];
        for (int i = 0; i < c->n; ++i) {
            int &m = map[(size_t)lab[i]];
            if (m == 0) m = ++next;
            dst[i] = m;
        }
    }
    const double ll = loglik_host(c, R.hi, R.ssize.data(), c->pinB[REC_SLOT]);                    // mcmc.jl:551
    const double lp = logprior_host(c, R.ssize.data(), R.slabel.data(), R.r, R.p);
    if (out->K) out->K[j] = R.K;
    if (out->r) out->r[j] = R.r;
    if (out->p) out->p[j] = R.p;
    if (out->loglik) out->loglik[j] = ll;
    if (out->logposterior) out->logposterior[j] = ll + lp;                                        // mcmc.jl:552
    R.active = false;
    return RC_OK;
}

}  // namespace chain

extern "C" int32_t rc_scalar_updates(uint64_t seed, uint64_t iter, double r, double p, const int64_t *sizes, int64_t K,
                                     int64_t n, double eta, double sigma, double proposalsd_r, double u, double v,
                                     double *r_out, double *p_out, uint8_t *accept_out)
{
    if (!sizes || !r_out || !p_out || !accept_out || K < 1 || n < K) return fail(nullptr, RC_ERR_ARG, "rc_scalar_updates: bad argument");
    if (!(r > 0.0) || !(p > 0.0 && p < 1.0)) return fail(nullptr, RC_ERR_ARG, "rc_scalar_updates: need r > 0 and 0 < p < 1");
    std::vector<int64_t> C(sizes, sizes + K);
    bool acc = false;
    *r_out = chain::sample_r(seed, iter, r, p, C, eta, sigma, proposalsd_r, &acc);
    *accept_out = acc;
    *p_out = chain::sample_p(seed, iter, K, n, *r_out, u, v);
    return RC_OK;
}

extern "C" int32_t rc_run_chain(rc_ctx *c, const rc_chain_options *o, rc_chain_outputs *out)
{
    if (!c || !o || !out) return fail(c, RC_ERR_ARG, "rc_run_chain: NULL argument");
    if (!c->have_params || !c->have_state) return fail(c, RC_ERR_STATE, "rc_run_chain: rc_set_params and rc_set_state must be called first");
    if (o->numiters < 0 || o->burnin < 0 || o->thin < 1 || o->numGibbs < 0 || o->numMH < 0)
        return fail(c, RC_ERR_ARG, "rc_run_chain: need numiters, burnin, numGibbs, numMH >= 0 and thin >= 1");
    if (!(o->r0 > 0.0) || !(o->p0 > 0.0 && o->p0 < 1.0)) return fail(c, RC_ERR_ARG, "rc_run_chain: need r0 > 0 and 0 < p0 < 1");
    if (!(o->proposalsd_r > 0.0)) return fail(c, RC_ERR_ARG, "rc_run_chain: need proposalsd_r > 0");
    if (o->numMH > 0 && (!c->hostD || !c->hostL)) return fail(c, RC_ERR_STATE, "rc_run_chain: numMH > 0 needs rc_attach_host_matrices");
    if ((o->r_trace == nullptr) != (o->p_trace == nullptr)) return fail(c, RC_ERR_ARG, "rc_run_chain: give both r_trace and p_trace or neither");
    HIPCHK(c, hipSetDevice(c->dev));
    const int n = c->n;
    const rc_params &P = c->P;
    int32_t rc = sync_and_check(c, true);
    if (rc != RC_OK) return rc;
    double r = o->r0, p = o->p0;
    std::vector<int64_t> C;
    int64_t j = 0;
    chain::Pending pend;
    bool pend_enqueued = false;
    const auto t0 = std::chrono::steady_clock::now();
    for (int64_t i = 1; i <= o->numiters; ++i) {
        const uint64_t it = o->first_iter + (uint64_t)(i - 1);
        if (!o->r_trace || pend.active) {
            // K and the sizes of the current state: host-mapped summary of the last sweep (or of rc_set_state)
            rc = sync_and_check(c);
            if (rc != RC_OK) return rc;
        }
        if (pend.active) {                      // the previous iteration ended in a state to record: device part now
            rc = chain::record_enqueue(c, pend, out->clusts != nullptr);
            if (rc != RC_OK) return rc;
            pend_enqueued = true;
            if (o->numMH > 0) {                 // proposals run their own loglik through the same staging: finish first
                rc = chain::record_finish(c, pend, out);
                if (rc != RC_OK) return rc;
                pend_enqueued = false;
            }
        }
        if (o->r_trace) {
            r = o->r_trace[i - 1]; p = o->p_trace[i - 1];
            if (!(r > 0.0) || !(p > 0.0 && p < 1.0)) return fail(c, RC_ERR_ARG, "rc_run_chain: r_trace/p_trace entry %lld out of range", (long long)i);
        } else {
            chain::sizes_by_label(c, C);
            bool acc = false;
            r = chain::sample_r(o->seed, it, r, p, C, P.eta, P.sigma, o->proposalsd_r, &acc);      // mcmc.jl:538
            if (out->r_acceptances) out->r_acceptances[i - 1] = acc;
            p = chain::sample_p(o->seed, it, (int64_t)C.size(), n, r, P.u, P.v);                  // mcmc.jl:539
        }
        if (out->r_all) out->r_all[i - 1] = r;
        if (out->p_all) out->p_all[i - 1] = p;
        bool accepted_any = false;
        if (o->numMH > 0) {                                                                       // mcmc.jl:372-474
            if (o->splitmerge_mode == RC_SM_AS_WRITTEN) {
                rc = rc_state_checkpoint(c);
                if (rc != RC_OK) return rc;
            }
            for (int64_t mh = 0; mh < o->numMH; ++mh) {
                uint8_t a = 0, s = 0;
                rc = rc_splitmerge(c, r, p, o->numGibbs, o->seed, it, (uint64_t)mh, &a, &s);
                if (rc != RC_OK) return rc;
                if (out->splitmerge_acceptances) out->splitmerge_acceptances[(i - 1) * o->numMH + mh] = a;
                if (out->splitmerge_splits) out->splitmerge_splits[(i - 1) * o->numMH + mh] = s;
                accepted_any |= a != 0;
            }
        }
        if (accepted_any && o->splitmerge_mode == RC_SM_AS_WRITTEN) {
            // Q1: `state = finalstate` (mcmc.jl:470) rebinds a local name — the accepted proposal and
            // the closing Gibbs scan (mcmc.jl:477) act on an object the caller never sees
            rc = rc_state_restore(c);
            if (rc != RC_OK) return rc;
        } else {
            rc = rc_gibbs_sweep_async(c, r, p, o->seed, it);                                      // mcmc.jl:477
            if (rc != RC_OK) return rc;
        }
        if (pend_enqueued) {                    // host part of the previous sample, under the sweep just launched
            rc = chain::record_finish(c, pend, out);
            if (rc != RC_OK) return rc;
            pend_enqueued = false;
        }
        if (i > o->burnin && (i - o->burnin) % o->thin == 0) {                                    // mcmc.jl:546
            if (j >= o->max_samples) return fail(c, RC_ERR_ARG, "rc_run_chain: more samples than max_samples=%lld", (long long)o->max_samples);
            pend.active = true;
            pend.j = j++;
            pend.r = r; pend.p = p;
        }
    }
    rc = sync_and_check(c);
    if (rc != RC_OK) return rc;
    if (pend.active) {
        rc = chain::record_enqueue(c, pend, out->clusts != nullptr);
        if (rc != RC_OK) return rc;
        rc = chain::record_finish(c, pend, out);
        if (rc != RC_OK) return rc;
        rc = sync_and_check(c);
        if (rc != RC_OK) return rc;
    }
    out->num_samples = j;
    out->runtime_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    out->r_final = r;
    out->p_final = p;
    return RC_OK;
}
