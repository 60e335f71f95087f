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

// chains running rc_run_chain in this process right now (rc_run_chains announces all of its chains before any starts)
static std::atomic<int> g_chains_running{0};
static thread_local bool t_counted_by_driver = false;

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
    int kcap = 0;             // slot capacity when the sample was taken: its slot ids are below this, whatever the capacity is when the host part runs (a wide context may have narrowed since)
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
    R.kcap = c->kcap;
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
        std::vector<int> map((size_t)std::max(R.kcap, c->kcap), 0);
        int next = 0;
        const unsigned short *lab = c->pinLab[REC_SLOT];
        for (int i = 0; i < c->n; ++i) {
            int &m = map[(size_t)lab[i]];
            if (m == 0) m = ++next;
            dst[i] = m;
        }
    }
    const double ll = loglik_host(c, R.hi, R.ssize.data(), c->pinB[REC_SLOT], R.slabel.data());                    // mcmc.jl:551
    const double lp = logprior_host(c, R.ssize.data(), R.slabel.data(), R.r, R.p, (int)R.ssize.size());   // (the sample's own slot count: the context may have narrowed since)
    if (out->K) out->K[j] = R.K;
    if (out->r) out->r[j] = R.r;
    if (out->p) out->p[j] = R.p;
    if (out->loglik) out->loglik[j] = ll;
    if (out->logposterior) out->logposterior[j] = ll + lp;                                        // mcmc.jl:552
    R.active = false;
    return RC_OK;
}

// ---- speculative pipeline for numMH > 0 --------------------------------------------------------------------------
// A split–merge proposal is 0.8 ms of sequential host work (the restricted scans, mcmc.jl:259-354) against a 0.09 ms
// sweep, and it is almost always rejected.  A rejected proposal leaves the state untouched, so the loop SPECULATES that
// every proposal of an iteration is rejected: it takes a snapshot of the state an iteration starts from (labels, slot
// tables, block sums), launches the sweep at once and hands the proposals to a worker thread, which decides them on the
// snapshot alone (proposal_core: no device, no context state).  Several iterations are in flight.  Iterations are
// confirmed in order; a sample is recorded only from confirmed iterations (labels and block sums of the following
// snapshot).  When a proposal turns out to be accepted — or is a split, whose likelihood needs the device — the loop rolls
// back: workers are drained, the device returns to the snapshot's labels (exact integer corrections, as
// rc_state_restore), that iteration is redone by the synchronous path, and speculation restarts behind it.  The
// speculation depth halves on a rollback and doubles again after 16 clean iterations, so a regime in which most
// proposals are splits degrades to the synchronous loop instead of thrashing.  Results are those of the synchronous loop
// bit for bit: every draw is a function of (seed, iteration, state) only.
struct SpecSlot {
    int64_t it = 0;                          // 1-based iteration whose initial state this is
    double r = 0, p = 0;
    int hi = 0, K = 0, capB = 0;
    std::vector<int> ssize, slabel;
    unsigned short *dev_row = nullptr, *pin_lab = nullptr;
    long long *pin_B = nullptr;
    hipEvent_t ev = nullptr;
    std::vector<int64_t> labels, sizes;      // by point, by label (the proposal's view of the state)
    std::vector<uint8_t> acc, spl;
    bool clean = false;                      // every proposal decided on the host and rejected (or skipped)
    bool split_pending = false;              // proposal number pend_mh is a split: the host part is in pend, the decision needs ll_fin
    int64_t pend_mh = 0;
    ProposalResult pend;
    int32_t err = RC_OK;
    const char *errmsg = nullptr;
    int state = 0;                           // 0 idle, 1 queued / running, 2 done   (guarded by SpecPool::m)
    int rec_busy = 0;                        // record jobs queued or running that read this snapshot (guarded by SpecPool::m)
};

// host part of a recorded sample (mcmc.jl:546-553), from the snapshot of the state the iteration ended in: pure function of the
// snapshot — run by a worker (the log-likelihood's K² scalar terms are 1-2 ms of long-double arithmetic when the chain moves)
static void spec_record_host(const rc_ctx *c, rc_ctx::LLCache &cache, const SpecSlot &after, int64_t j, double r, double p, rc_chain_outputs *out)
{
    const int nslots = (int)after.ssize.size();
    if (out->clusts) {   // sortlabels (utils.jl:69-74): relabel by order of first appearance
        int64_t *dst = out->clusts + (size_t)j * c->n;
        std::vector<int> map((size_t)nslots, 0);
        int next = 0;
        for (int i = 0; i < c->n; ++i) {
            int &m = map[(size_t)after.pin_lab[i]];
            if (m == 0) m = ++next;
            dst[i] = m;
        }
    }
    const double ll = loglik_host_c(c, cache, after.hi, after.ssize.data(), after.pin_B, after.slabel.data());         // mcmc.jl:551
    const double lp = logprior_host(c, after.ssize.data(), after.slabel.data(), r, p, nslots);
    if (out->K) out->K[j] = after.K;
    if (out->r) out->r[j] = r;
    if (out->p) out->p[j] = p;
    if (out->loglik) out->loglik[j] = ll;
    if (out->logposterior) out->logposterior[j] = ll + lp;                                       // mcmc.jl:552
}

struct SpecPool {
    rc_ctx *c;
    const rc_chain_options *o;
    std::vector<SpecSlot> slots;
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::deque<int> queue;
    struct RecJob { int si; int64_t j; double r, p; };
    std::deque<RecJob> rec_queue;            // recorded samples whose host part is still to be computed (confirmed iterations: never cancelled)
    int rec_open = 0;                        // record jobs queued or running
    rc_chain_outputs *out = nullptr;
    bool stop = false;

    SpecPool(rc_ctx *c_, const rc_chain_options *o_, int nslots) : c(c_), o(o_), slots((size_t)nslots) {}
    void start(int nworkers)
    {
        for (int w = 0; w < nworkers; ++w) threads.emplace_back([this] { run(); });
    }
    void run()
    {
        rc_ctx::LLCache cache;
        (void)hipSetDevice(c->dev);          // (the worker waits for snapshot events)
        for (;;) {
            int si = -1;
            RecJob rj{-1, 0, 0, 0};
            {
                std::unique_lock<std::mutex> lk(m);
                cv_work.wait(lk, [&] { return stop || !queue.empty() || !rec_queue.empty(); });
                if (stop) return;
                if (!rec_queue.empty()) { rj = rec_queue.front(); rec_queue.pop_front(); }
                else { si = queue.front(); queue.pop_front(); }
            }
            if (rj.si >= 0) {
                SpecSlot &a = slots[(size_t)rj.si];
                if (a.ev) (void)hipEventSynchronize(a.ev);     // the snapshot's copies have arrived
                spec_record_host(c, cache, a, rj.j, rj.r, rj.p, out);
                {
                    std::lock_guard<std::mutex> lk(m);
                    --a.rec_busy; --rec_open;
                }
                cv_done.notify_all();
                continue;
            }
            SpecSlot &s = slots[(size_t)si];
            if (s.ev) (void)hipEventSynchronize(s.ev);         // the snapshot's copies have arrived (the main thread did not wait for them)
            // the proposal's view of the snapshot — labels by point, sizes by label — is built here, not by the chain's main thread
            // (15 µs per iteration at n = 8192 on the path between two sweeps)
            {
                const int n = c->n;
                s.labels.resize((size_t)n); s.sizes.assign((size_t)n, 0);
                for (int q = 0; q < n; ++q) s.labels[(size_t)q] = s.slabel[(size_t)s.pin_lab[q]];
                for (int k = 0; k < (int)s.slabel.size(); ++k)
                    if (s.slabel[(size_t)k] > 0) s.sizes[(size_t)s.slabel[(size_t)k] - 1] = s.ssize[(size_t)k];
                s.acc.assign((size_t)o->numMH, 0); s.spl.assign((size_t)o->numMH, 0);
            }
            bool clean = true;
            for (int64_t mh = 0; mh < o->numMH && clean; ++mh) {
                ProposalSnapshot S0{s.labels.data(), s.sizes.data(), (int64_t)s.K, s.hi, s.ssize.data(), s.slabel.data(), s.pin_B};
                ProposalResult R;
                proposal_core(c, cache, S0, s.r, s.p, o->numGibbs, o->seed, o->first_iter + (uint64_t)(s.it - 1), (uint64_t)mh, R, false);
                if (R.err != RC_OK) { s.err = R.err; s.errmsg = R.errmsg; clean = false; break; }
                s.acc[(size_t)mh] = 0; s.spl[(size_t)mh] = R.split ? 1 : 0;
                if (R.needs_device) { s.split_pending = true; s.pend_mh = mh; s.pend = std::move(R); clean = false; }
                else if (R.accept) clean = false;
            }
            s.clean = clean;
            {
                std::lock_guard<std::mutex> lk(m);
                s.state = 2;
            }
            cv_done.notify_all();
        }
    }
    void submit(int si)
    {
        {
            std::lock_guard<std::mutex> lk(m);
            slots[(size_t)si].state = 1;
            queue.push_back(si);
        }
        cv_work.notify_one();
    }
    void submit_record(int si, int64_t j, double r, double p)
    {
        {
            std::lock_guard<std::mutex> lk(m);
            ++slots[(size_t)si].rec_busy; ++rec_open;
            rec_queue.push_back(RecJob{si, j, r, p});
        }
        cv_work.notify_one();
    }
    void wait_records(int si) { std::unique_lock<std::mutex> lk(m); cv_done.wait(lk, [&] { return slots[(size_t)si].rec_busy == 0; }); }   // before the slot's snapshot is overwritten
    void wait_all_records() { std::unique_lock<std::mutex> lk(m); cv_done.wait(lk, [&] { return rec_open == 0; }); }
    bool done(int si) { std::lock_guard<std::mutex> lk(m); return slots[(size_t)si].state == 2; }
    void wait(int si) { std::unique_lock<std::mutex> lk(m); cv_done.wait(lk, [&] { return slots[(size_t)si].state == 2; }); }
    void drain()   // nothing queued or running afterwards; every slot idle
    {
        std::unique_lock<std::mutex> lk(m);
        for (int si : queue) slots[(size_t)si].state = 0;      // never started
        queue.clear();
        cv_done.wait(lk, [&] { for (auto &s : slots) if (s.state == 1) return false; return true; });   // running ones finish
        for (auto &s : slots) s.state = 0;
    }
    ~SpecPool()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv_work.notify_all();
        for (auto &t : threads) t.join();
        if (c->sC) (void)hipStreamSynchronize(c->sC);   // the slots' events and pinned buffers go: no copy may be in flight,
        c->ev_blocks_busy = nullptr;                    // and nobody may wait on an event of theirs afterwards
        for (auto &s : slots) {
            if (s.dev_row) (void)hipFree(s.dev_row);
            if (s.pin_lab) (void)hipHostFree(s.pin_lab);
            if (s.pin_B) (void)hipHostFree(s.pin_B);
            if (s.ev) (void)hipEventDestroy(s.ev);
        }
    }
};

// snapshot of the state on the device (complete: the caller has synchronised) into slot s, asynchronously on stream A
// (prev != nullptr: the sweep that produced this state changed no label and prev is the snapshot of the state before it —
// the two states are identical, so the host copies are duplicated and only the device label row is copied, with no wait)
static int32_t spec_snapshot(rc_ctx *c, SpecSlot &s, const SpecSlot *prev, bool *need_wait)
{
    *need_wait = true;
    if (prev && prev->pin_B && prev->dev_row && s.dev_row && s.capB >= prev->hi) {
        if (prev->ev) HIPCHK(c, hipEventSynchronize(prev->ev));   // prev's host copies are complete (nobody else has waited for them)
        s.hi = prev->hi; s.K = prev->K; s.ssize = prev->ssize; s.slabel = prev->slabel;
        std::memcpy(s.pin_lab, prev->pin_lab, (size_t)c->n * sizeof(unsigned short));
        std::memcpy(s.pin_B, prev->pin_B, (size_t)prev->hi * prev->hi * 4 * sizeof(long long));
        HIPCHK(c, hipMemcpyAsync(s.dev_row, prev->dev_row, (size_t)c->ldc * sizeof(unsigned short), hipMemcpyDeviceToDevice, c->sA));
        *need_wait = false;
        return RC_OK;
    }
    s.hi = std::max(1, std::min(c->kcap, c->hsum->slot_hi));
    s.K = c->hsum->K;
    s.ssize.resize((size_t)c->kcap); s.slabel.resize((size_t)c->kcap);
    for (int k = 0; k < c->kcap; ++k) { s.ssize[(size_t)k] = c->hsum->size_label[2 * k]; s.slabel[(size_t)k] = c->hsum->size_label[2 * k + 1]; }
    if (!s.dev_row) {
        HIPCHK(c, hipMalloc((void **)&s.dev_row, (size_t)c->ldc * sizeof(unsigned short)));
        HIPCHK(c, hipHostMalloc((void **)&s.pin_lab, (size_t)(c->n + 8) * sizeof(unsigned short), hipHostMallocDefault));
        HIPCHK(c, hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
    }
    if (s.hi > s.capB) {
        if (s.pin_B) (void)hipHostFree(s.pin_B);
        s.pin_B = nullptr;
        s.capB = std::min(c->kcap, std::max(64, 2 * s.hi));
        HIPCHK(c, hipHostMalloc((void **)&s.pin_B, (size_t)s.capB * s.capB * 4 * sizeof(long long), hipHostMallocDefault));
    }
    int32_t rc = order_A_after_sweeps(c);
    if (rc != RC_OK) return rc;
    // Stream A — where the sweeps of the incremental mode run, and what the next resolver waits for in full mode — carries only
    // the two kernels that read the live state; the copies to the host (16 KB of labels, hi² x 32 B of block sums: 2 MB at 256
    // slots, 60-80 µs over PCIe) follow on the copy stream, beside the next sweep instead of in front of it.
    k_snapshot<<<(c->ldc + 255) / 256, 256, 0, c->sA>>>(c->slot_of, c->pi, c->n, c->ldc, s.dev_row);
    HIPCHK(c, hipGetLastError());
    rc = loglik_enqueue(c, s.hi, s.pin_B, c->sC);      // (records ev_k behind k_blocksums on stream A and makes stream C wait for it)
    if (rc != RC_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(s.pin_lab, s.dev_row, (size_t)c->n * sizeof(unsigned short), hipMemcpyDeviceToHost, c->sC));
    HIPCHK(c, hipEventRecord(s.ev, c->sC));
    c->ev_blocks_busy = s.ev;
    return RC_OK;
}

// the recorded sample of an iteration (mcmc.jl:546-553) from the snapshot of the state it ended in: the device part here (the label
// row into the co-clustering queue), the host part as a job of the pool (spec_record_host)
static int32_t spec_record(rc_ctx *c, SpecPool &pool, int after_si, int64_t j, double r, double p)
{
    SpecSlot &after = pool.slots[(size_t)after_si];
    int32_t rc = ensure_counts(c);
    if (rc != RC_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(c->snap + (size_t)c->snap_cnt * c->ldc, after.dev_row, (size_t)c->ldc * sizeof(unsigned short),
                             hipMemcpyDeviceToDevice, c->sA));
    if (++c->snap_cnt == RC_CC_BATCH) {
        rc = flush_counts(c);
        if (rc != RC_OK) return rc;
    }
    pool.submit_record(after_si, j, r, p);
    return RC_OK;
}

// Log-likelihood of the state a SPLIT proposal leads to, evaluated against the snapshot the proposal was made on and
// without touching the live device state: the block sums of the new cluster c0 (the points labelled R.cfinal[i] ≠ their
// snapshot label) against every cluster come from k_split_eval on the matrices, those of the remainder c1 by exact
// subtraction from the snapshot's block sums.  The new cluster takes the lowest free slot of the snapshot's table, as
// apply_labels would give it, so the terms are summed in the same order and the value is the one the synchronous path
// computes after applying the proposal.  Returns RC_ERR_CAPACITY (no error text) when there is no free slot: the caller
// falls back to the synchronous path.
struct SplitScratch {
    unsigned short *d_bucket = nullptr; int *d_rows = nullptr; long long *d_out = nullptr, *h_out = nullptr;
    int cap = 0;                 // slot capacity d_out / h_out were sized for (the context's capacity can grow)
    std::vector<unsigned short> bucket; std::vector<int> rows;
    void release_out() { if (d_out) (void)hipFree(d_out); if (h_out) (void)hipHostFree(h_out); d_out = nullptr; h_out = nullptr; }
    ~SplitScratch() { if (d_bucket) (void)hipFree(d_bucket); if (d_rows) (void)hipFree(d_rows); release_out(); }   // every early return of the loop frees it
};

static int32_t spec_eval_split(rc_ctx *c, rc_ctx::LLCache &cache, SplitScratch &X, const SpecSlot &s, ProposalResult &R)
{
    const int n = c->n, hi = s.hi;
    int f = -1;
    for (int k = 0; k < (int)s.slabel.size(); ++k)     // (the snapshot's own table: the context's capacity may have grown since)
        if (s.slabel[(size_t)k] == 0) { f = k; break; }
    if (f < 0) return RC_ERR_CAPACITY;
    const int h2 = std::max(hi, f + 1);
    if ((size_t)h2 * 4 * sizeof(u64) > 48 * 1024) return RC_ERR_CAPACITY;   // LDS bins of k_split_eval (beyond: synchronous path)
    if (!X.d_bucket) {
        HIPCHK(c, hipMalloc((void **)&X.d_bucket, (size_t)c->ld * sizeof(unsigned short)));
        HIPCHK(c, hipMalloc((void **)&X.d_rows, (size_t)n * sizeof(int)));
        X.bucket.resize((size_t)c->ld); X.rows.resize((size_t)n);
    }
    if (X.cap < c->kcap) {
        X.release_out();
        X.cap = c->kcap;
        HIPCHK(c, hipMalloc((void **)&X.d_out, (size_t)(X.cap + 1) * 4 * sizeof(long long)));
        HIPCHK(c, hipHostMalloc((void **)&X.h_out, (size_t)(X.cap + 1) * 4 * sizeof(long long), hipHostMallocDefault));
    }
    // buckets in the internal point order: the snapshot's slot, except that the points moved to the new label go to f
    int si = -1, nrows = 0, new_label = 0;
    std::fill(X.bucket.begin(), X.bucket.end(), (unsigned short)0);
    for (int q = 0; q < n; ++q) {
        const int u = c->h_pi[(size_t)q];
        int b = s.pin_lab[q];
        if (R.cfinal[(size_t)q] != s.labels[(size_t)q]) { si = b; b = f; X.rows[(size_t)nrows++] = u; new_label = (int)R.cfinal[(size_t)q]; }
        X.bucket[(size_t)u] = (unsigned short)b;
    }
    if (nrows == 0 || si < 0) return fail(c, RC_ERR_STATE, "split evaluation: the proposal moves no point");
    HIPCHK(c, hipMemcpyAsync(X.d_bucket, X.bucket.data(), (size_t)c->ld * sizeof(unsigned short), hipMemcpyHostToDevice, c->sA));
    HIPCHK(c, hipMemcpyAsync(X.d_rows, X.rows.data(), (size_t)nrows * sizeof(int), hipMemcpyHostToDevice, c->sA));
    HIPCHK(c, hipMemsetAsync(X.d_out, 0, (size_t)h2 * 4 * sizeof(long long), c->sA));
    View V = make_view(c);
    k_split_eval<<<nrows, 256, (size_t)h2 * 4 * sizeof(u64), c->sA>>>(V, X.d_bucket, X.d_rows, h2, X.d_out);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(X.h_out, X.d_out, (size_t)h2 * 4 * sizeof(long long), hipMemcpyDeviceToHost, c->sA));
    HIPCHK(c, hipStreamSynchronize(c->sA));
    // proposed block sums (h2 × h2 × 4) and sizes
    std::vector<long long> B2((size_t)h2 * h2 * 4, 0);
    std::vector<int> sz2((size_t)h2, 0);
    for (int t = 0; t < hi; ++t) {
        sz2[(size_t)t] = s.ssize[(size_t)t];
        for (int k = 0; k < hi; ++k)
            for (int w = 0; w < 4; ++w) B2[((size_t)t * h2 + k) * 4 + w] = s.pin_B[((size_t)t * hi + k) * 4 + w];
    }
    const long long *T = X.h_out;
    auto at = [&](int t, int k, int w) -> long long & { return B2[((size_t)t * h2 + k) * 4 + w]; };
    for (int w = 0; w < 4; ++w) {
        for (int k = 0; k < h2; ++k) {
            if (k == f || k == si) continue;
            at(f, k, w) = T[k * 4 + w]; at(k, f, w) = T[k * 4 + w];
            at(si, k, w) -= T[k * 4 + w]; at(k, si, w) -= T[k * 4 + w];
        }
        const long long t00 = T[f * 4 + w], t01 = T[si * 4 + w];
        at(si, si, w) -= t00 + 2 * t01;
        at(f, f, w) = t00;
        at(f, si, w) = t01; at(si, f, w) = t01;
    }
    sz2[(size_t)f] = nrows;
    sz2[(size_t)si] -= nrows;
    std::vector<int> lab2(s.slabel.begin(), s.slabel.begin() + std::min<size_t>(s.slabel.size(), (size_t)h2));
    lab2.resize((size_t)h2, 0);
    lab2[(size_t)f] = new_label;
    R.ll_fin = loglik_host_c(c, cache, h2, sz2.data(), B2.data(), lab2.data());
    return RC_OK;
}

static int32_t run_chain_speculative(rc_ctx *c, const rc_chain_options *o, rc_chain_outputs *out)
{
    const int n = c->n;
    const rc_params &P = c->P;
    // iterations in flight: a proposal is 0.8 ms of a worker's time, an iteration 70-100 µs of the main thread's — a dozen
    // iterations deep the main thread waited for the oldest job
    int Dmax = std::max(1, std::min(64, c->opt_chain_depth > 0 ? c->opt_chain_depth : 24));
    // every iteration in flight holds a snapshot of the K x K block sums in pinned memory (32 B per pair: 2 MB at 256 slots, 134 MB at
    // 2048): a chain that starts among many hundreds of clusters keeps fewer in flight
    if (c->opt_chain_depth <= 0 && c->hsum->slot_hi > 512) Dmax = std::min(Dmax, c->hsum->slot_hi > 1024 ? 3 : 8);
    // worker threads: the host's cores shared by the chains this process runs at once (rc_run_chains: one per GPU — eight
    // chains must not start 200 threads), one core left to each chain's main thread; at most 24
    const int chains_here = std::max(1, g_chains_running.load());
    const int cores = (int)std::max(2u, std::thread::hardware_concurrency());
    int nw = c->opt_chain_workers > 0 ? c->opt_chain_workers : std::min(24, std::max(1, cores / chains_here - 1));
    nw = std::max(1, std::min(nw, Dmax));
    const long long grows0 = c->n_grows;
    const int Rn = Dmax + 2;
    SpecPool pool(c, o, Rn);
    pool.out = out;
    pool.start(nw);
    auto slot_of_it = [&](int64_t it) -> int { return (int)(it % Rn); };
    int32_t rc = sync_and_check(c, true);
    if (rc != RC_OK) return rc;
    double r_prev = o->r0, p_prev = o->p0;   // r, p of the last started iteration
    std::vector<int64_t> C;
    int64_t j = 0, conf = 1, i = 1, prev_it = -1;   // prev_it: iteration of the last snapshot taken with the device untouched since
    int D = Dmax, clean_run = 0;
    SplitScratch scratch;
    // A device error (slot capacity beyond the library's maximum, barrier time-out) of a sweep that was launched on speculation
    // belongs to an iteration that may yet be voided by an accepted proposal before it: it is kept here, no further iteration
    // is started, and it is reported only once every iteration before the failed sweep has been confirmed clean — what the
    // synchronous loop would have reported too.  A rollback before that point discards it with the void sweeps.
    int32_t spec_err = RC_OK;
    std::string spec_msg;
    long long n_rollbacks = 0, n_split_evals = 0; double t_rollback = 0, t_sync = 0, t_snap = 0, t_evwait = 0, t_build = 0, t_confirm = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const int64_t N = o->numiters;
    const auto t0 = std::chrono::steady_clock::now();
    auto recording = [&](int64_t it) { return it > o->burnin && (it - o->burnin) % o->thin == 0; };
    while (conf <= N) {
        // 1. start iteration i (i == N + 1: only the closing snapshot, the state the last iteration ended in)
        if (i <= N + 1 && i - conf <= D && spec_err == RC_OK) {
            const auto ta = now();
            rc = sync_and_check(c);                        // the state iteration i starts from is complete; K and sizes visible
            if (rc != RC_OK) {
                if (i == conf) return rc;                  // every iteration before i is confirmed: the sweep that failed is real
                spec_err = rc; spec_msg = c->err;
                continue;
            }
            const auto tb = now(); t_sync += secs(ta, tb);
            SpecSlot &s = pool.slots[(size_t)slot_of_it(i)];
            pool.wait_records(slot_of_it(i));              // (a recorded sample may still be read from the snapshot this slot held)
            s.it = i;
            if (i <= N) {
                const uint64_t it = o->first_iter + (uint64_t)(i - 1);
                double r = r_prev, p = p_prev;
                if (o->r_trace) {
                    r = o->r_trace[i - 1]; p = o->p_trace[i - 1];
                    if (!(r > 0.0) || !(p > 0.0 && p < 1.0)) return fail(c, RC_ERR_ARG, "rc_run_chain: r_trace/p_trace entry %lld out of range", (long long)i);
                } else {
                    sizes_by_label(c, C);
                    bool acc = false;
                    r = sample_r(o->seed, it, r, p, C, P.eta, P.sigma, o->proposalsd_r, &acc);            // mcmc.jl:538
                    if (out->r_acceptances) out->r_acceptances[i - 1] = acc;
                    p = sample_p(o->seed, it, (int64_t)C.size(), n, r, P.u, P.v);                        // mcmc.jl:539
                }
                if (out->r_all) out->r_all[i - 1] = r;
                if (out->p_all) out->p_all[i - 1] = p;
                s.r = r; s.p = p;
                r_prev = r; p_prev = p;
            }
            // without proposals a snapshot is needed only as the state a recorded iteration ended in
            const bool need_snap = o->numMH > 0 || (i >= 2 && recording(i - 1));
            if (!need_snap) {
                if (i <= N) {
                    rc = rc_gibbs_sweep_async(c, s.r, s.p, o->seed, o->first_iter + (uint64_t)(i - 1));     // mcmc.jl:477
                    if (rc != RC_OK) return rc;
                }
                { std::lock_guard<std::mutex> lk(pool.m); s.clean = true; s.err = RC_OK; s.split_pending = false; s.state = 2; }
                t_snap += secs(tb, now());
                ++i;
                continue;
            }
            // an unchanged state (the last sweep moved no label, nothing else touched the device) shares the previous snapshot
            const SpecSlot *prev = (prev_it == i - 1 && c->hsum->n_changes == 0) ? &pool.slots[(size_t)slot_of_it(i - 1)] : nullptr;
            bool need_wait = true;
            // (a shared snapshot does not read the device state: the sweep goes first, the host copies are made under it)
            const bool sweep_first = prev && prev->pin_B && prev->dev_row && s.dev_row && s.capB >= prev->hi;
            if (sweep_first && i <= N) {
                rc = rc_gibbs_sweep_async(c, s.r, s.p, o->seed, o->first_iter + (uint64_t)(i - 1));     // mcmc.jl:477, speculatively
                if (rc != RC_OK) return rc;
            }
            rc = spec_snapshot(c, s, prev, &need_wait);
            if (rc != RC_OK) return rc;
            prev_it = i;
            if (!sweep_first && i <= N) {
                rc = rc_gibbs_sweep_async(c, s.r, s.p, o->seed, o->first_iter + (uint64_t)(i - 1));     // mcmc.jl:477, speculatively
                if (rc != RC_OK) return rc;
            }
            const auto tc = now(); t_snap += secs(tb, tc);
            (void)need_wait;                               // (the worker that takes the job waits for the snapshot's copies)
            const auto td = now(); t_evwait += secs(tc, td);
            if (i <= N && o->numMH > 0) {
                s.clean = false; s.err = RC_OK; s.split_pending = false;
                pool.submit(slot_of_it(i));      // (the worker builds the job's label and size vectors from the snapshot)
            } else if (i <= N) {
                std::lock_guard<std::mutex> lk(pool.m);     // no proposals: nothing to decide
                s.clean = true; s.err = RC_OK; s.split_pending = false; s.state = 2;
            }
            t_build += secs(td, now());
            ++i;
        }
        // 2. confirm finished iterations in order
        while (conf <= N && conf < i) {
            const int si = slot_of_it(conf);
            SpecSlot &s = pool.slots[(size_t)si];
            const bool can_start_more = (spec_err == RC_OK && i <= N + 1 && i - conf <= D);
            if (!pool.done(si)) {
                if (can_start_more) break;
                pool.wait(si);
            }
            if (s.err != RC_OK) return fail(c, s.err, "%s", s.errmsg ? s.errmsg : "split-merge proposal failed");
            if (s.clean) {
                // iteration conf is confirmed, so its sweep was real; if that is the sweep whose failure is on hold, report it now
                if (spec_err != RC_OK && conf + 1 >= i) return fail(c, spec_err, "%s", spec_msg.c_str());
                if (recording(conf) && conf + 1 >= i) break;                  // the snapshot this sample is read from comes next
                for (int64_t mh = 0; mh < o->numMH; ++mh) {
                    if (out->splitmerge_acceptances) out->splitmerge_acceptances[(conf - 1) * o->numMH + mh] = 0;
                    if (out->splitmerge_splits) out->splitmerge_splits[(conf - 1) * o->numMH + mh] = s.spl[(size_t)mh];
                }
                if (recording(conf)) {
                    if (j >= o->max_samples) return fail(c, RC_ERR_ARG, "rc_run_chain: more samples than max_samples=%lld", (long long)o->max_samples);
                    rc = spec_record(c, pool, slot_of_it(conf + 1), j++, s.r, s.p);
                    if (rc != RC_OK) return rc;
                }
                { std::lock_guard<std::mutex> lk(pool.m); s.state = 0; }
                ++conf;
                if (++clean_run >= 16 && D < Dmax) { D = std::min(Dmax, 2 * D); clean_run = 0; }
                continue;
            }
            if (s.split_pending) {
                // split proposals: the worker did the scans; the likelihood of the proposed state is evaluated here against the
                // snapshot, off the live state.  A rejected split (the usual case) costs no rollback.
                bool resolved = true;
                int64_t mh = s.pend_mh;
                ProposalResult *R = &s.pend;
                ProposalResult Rtmp;
                for (;;) {
                    if (R->needs_device) {
                        const int32_t erc = spec_eval_split(c, c->llc, scratch, s, *R);
                        if (erc == RC_ERR_CAPACITY) { resolved = false; break; }
                        if (erc != RC_OK) return erc;
                        ++n_split_evals;
                        if (finish_decision(*R, o->seed, o->first_iter + (uint64_t)(conf - 1), (uint64_t)mh)) { resolved = false; break; }
                    } else if (R->accept) { resolved = false; break; }
                    s.acc[(size_t)mh] = 0; s.spl[(size_t)mh] = R->split ? 1 : 0;
                    if (++mh >= o->numMH) break;
                    // the remaining proposals of the iteration see the same (unchanged) state
                    ProposalSnapshot S0{s.labels.data(), s.sizes.data(), (int64_t)s.K, s.hi, s.ssize.data(), s.slabel.data(), s.pin_B};
                    Rtmp = ProposalResult();
                    proposal_core(c, c->llc, S0, s.r, s.p, o->numGibbs, o->seed, o->first_iter + (uint64_t)(conf - 1), (uint64_t)mh, Rtmp, false);
                    if (Rtmp.err != RC_OK) return fail(c, Rtmp.err, "%s", Rtmp.errmsg);
                    R = &Rtmp;
                }
                s.split_pending = false;
                if (resolved) { s.clean = true; continue; }          // confirmed by the clean branch on the next pass
            }
            // rollback: a proposal of iteration `conf` is accepted (or could not be evaluated off-line).  Everything started after it is void.
            const auto tr0 = now(); ++n_rollbacks;
            pool.drain();
            c->checkpoint = s.labels;                                          // the state iteration conf started from
            if (spec_err != RC_OK) {
                // the failed sweep came after this iteration and is void with it: rc_set_state installs the snapshot's labels
                // afresh (the device error word and the half-swept state go with it)
                spec_err = RC_OK; spec_msg.clear();
                rc = rc_set_state(c, s.labels.data());
                if (rc != RC_OK) return rc;
            } else {
                rc = sync_and_check(c, true);
                if (rc != RC_OK) {
                    // a speculative sweep behind this iteration failed after the last check: void as well
                    rc = rc_set_state(c, s.labels.data());
                    if (rc != RC_OK) return rc;
                } else {
                    rc = rc_state_restore(c);
                    if (rc != RC_OK) return rc;
                }
            }
            const uint64_t it = o->first_iter + (uint64_t)(conf - 1);
            bool accepted_any = false;
            for (int64_t mh = 0; mh < o->numMH; ++mh) {                        // the synchronous path, mcmc.jl:372-474
                uint8_t a = 0, sp = 0;
                rc = rc_splitmerge(c, s.r, s.p, o->numGibbs, o->seed, it, (uint64_t)mh, &a, &sp);
                if (rc != RC_OK) return rc;
                if (out->splitmerge_acceptances) out->splitmerge_acceptances[(conf - 1) * o->numMH + mh] = a;
                if (out->splitmerge_splits) out->splitmerge_splits[(conf - 1) * o->numMH + mh] = sp;
                accepted_any |= a != 0;
            }
            if (accepted_any && o->splitmerge_mode == RC_SM_AS_WRITTEN) {
                rc = rc_state_restore(c);                                      // Q1: the caller never sees the accepted proposal
                if (rc != RC_OK) return rc;
            } else {
                rc = rc_gibbs_sweep_async(c, s.r, s.p, o->seed, it);
                if (rc != RC_OK) return rc;
            }
            rc = sync_and_check(c);
            if (rc != RC_OK) return rc;
            if (recording(conf)) {
                if (j >= o->max_samples) return fail(c, RC_ERR_ARG, "rc_run_chain: more samples than max_samples=%lld", (long long)o->max_samples);
                Pending pend;
                pend.active = true; pend.j = j++; pend.r = s.r; pend.p = s.p;
                rc = record_enqueue(c, pend, out->clusts != nullptr);
                if (rc != RC_OK) return rc;
                rc = record_finish(c, pend, out);
                if (rc != RC_OK) return rc;
            }
            r_prev = s.r; p_prev = s.p;
            ++conf;
            i = conf;
            prev_it = -1;
            D = std::max(1, D / 2); clean_run = 0;
            t_rollback += secs(tr0, now());
        }
    }
    pool.drain();
    pool.wait_all_records();
    rc = sync_and_check(c, true);
    if (rc != RC_OK) return rc;
    out->num_samples = j;
    out->runtime_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (c->sm_profile)
        fprintf(stderr, "[rc_run_chain speculative] %lld iterations %.3f s: %lld rollbacks %.3f s, %lld splits evaluated off-line; wait for sweep %.3f s, snapshot + launch %.3f s, wait for snapshot %.3f s, job build %.3f s; %d workers depth %d\n",
                (long long)N, out->runtime_s, n_rollbacks, t_rollback, n_split_evals, t_sync, t_snap, t_evwait, t_build, nw, Dmax);
    (void)t_confirm;
    c->chain_rollbacks = n_rollbacks; c->chain_split_evals = n_split_evals; c->chain_workers = nw; c->chain_grows = c->n_grows - grows0;
    out->r_final = r_prev;
    out->p_final = p_prev;
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

extern "C" int32_t rc_chain_stats(rc_ctx *c, int64_t *rollbacks, int64_t *split_evals, int64_t *workers, int64_t *grows)
{
    if (!c) return fail(c, RC_ERR_ARG, "rc_chain_stats: NULL ctx");
    if (rollbacks) *rollbacks = c->chain_rollbacks;
    if (split_evals) *split_evals = c->chain_split_evals;
    if (workers) *workers = c->chain_workers;
    if (grows) *grows = c->chain_grows;
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
    c->chain_rollbacks = c->chain_split_evals = c->chain_workers = c->chain_grows = 0;
    // (a wide context may narrow under the synchronous loop below — sweep_enqueue: a sample's deferred host part keeps the slot count it
    // was taken with, chain::Pending::kcap; the pipelined loop never sees a wide context)
    struct Running {   // counted while the loop runs (unless rc_run_chains has counted all of its chains already)
        bool counted;
        Running() : counted(!t_counted_by_driver) { if (counted) ++g_chains_running; }
        ~Running() { if (counted) --g_chains_running; }
    } running;
    // The pipelined loop also runs chains without split-merge proposals: its snapshots are then taken only for the iterations that
    // are recorded, and the host part of a recorded sample is a job of its worker pool instead of the main thread's (thin = 1, the
    // reference's default, at N = 8192: 8.4 k -> 11 k it/s stationary, 1.4 k -> 3 k while 40 labels move per sweep).  The loop below
    // is the synchronous form (RC_CHAIN_PIPELINE=0): same chain, bit for bit.
    // (a wide context — thousands of clusters — runs the synchronous form: every snapshot of the pipelined loop holds K x K block sums)
    if (o->numiters > 0 && c->opt_chain_pipeline && !c->wide)
        return chain::run_chain_speculative(c, o, out);
    const long long grows0 = c->n_grows;
    struct GrowNote { rc_ctx *c; long long g0; ~GrowNote() { c->chain_grows = c->n_grows - g0; } } grow_note{c, grows0};
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
