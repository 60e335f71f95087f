"""Independent NumPy/SciPy transcription of the reference's Gibbs-sweep path (TEST INFRASTRUCTURE).

Second restatement, written separately from oracle/rc_oracle.c, used (a) by tests/golden/make_golden.py
to produce the committed golden vectors and (b) by tests to cross-check the C oracle.  It follows the
reference's formulas literally (scipy.special.gammaln for loggamma, f64 sums):

  sample_labels_Gibbs!   /root/reference/src/mcmc.jl:158-256
  loglik                 /root/reference/src/mcmc.jl:1-56
  logprior               /root/reference/src/mcmc.jl:58-78
  sample_logweights      /root/reference/src/utils.jl:2-6
  sortlabels             /root/reference/src/utils.jl:69-74
  adjacencymatrix        /root/reference/src/utils.jl:59-63
  MCMCData / MCMCState   /root/reference/src/types.jl:131-157
  likelihood hyperparameters from a labelling   /root/reference/src/prior.jl:73-75,96-110

The uniform stream is the counter-based Philox4x32-10 stream defined in DESIGN.md (Julia's RNG stream
cannot be reproduced outside Julia).
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import gammaln

M32 = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    c0, c1, c2, c3 = [int(x) & M32 for x in ctr]
    k0, k1 = [int(x) & M32 for x in key]
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


def uniform(seed: int, sweep: int, i: int, pos: int) -> float:
    """u(seed, sweep, i, pos) in (0,1): 52 random bits + 0.5, scaled by 2^-52."""
    c = philox4x32_10((pos, i, sweep & M32, (sweep >> 32) & M32), (seed & M32, (seed >> 32) & M32))
    bits = ((c[0] << 32) | c[1]) >> 12
    return (bits + 0.5) * 2.0 ** -52


def make_logD(D: np.ndarray) -> np.ndarray:
    """types.jl:155 — log.(D - Diagonal(D) + I)."""
    if np.any(D != D.T):
        raise ValueError("D must be symmetric.")
    M = D - np.diag(np.diag(D)) + np.eye(D.shape[0])
    with np.errstate(divide="ignore"):
        return np.log(M)


def state_from_labels(clusts: np.ndarray):
    n = len(clusts)
    sizes = np.bincount(clusts, minlength=n + 1)[1:].astype(np.int64)  # counts(clusts, 1:n)
    return sizes, int(np.sum(sizes > 0))


def likelihood_hyperparams(D: np.ndarray, labels: np.ndarray) -> dict:
    """prior.jl:73-75,96-110 with the notional clustering replaced by the given labels: A/B = within/
    between-cluster upper-triangle distances, δ = Gamma-MLE shape, α=|A|δ1, β=ΣA, ζ=|B|δ2, γ=ΣB."""
    n = D.shape[0]
    iu = np.triu_indices(n, 1)
    same = labels[iu[0]] == labels[iu[1]]
    A = D[iu][same]
    B = D[iu][~same]

    def gamma_shape_mle(x):
        # Newton iteration on log(k) - digamma(k) = log(mean) - mean(log), Distributions.fit_mle(Gamma)
        from scipy.special import digamma, polygamma
        s = np.log(x.mean()) - np.log(x).mean()
        k = (3 - s + np.sqrt((s - 3) ** 2 + 24 * s)) / (12 * s)
        for _ in range(100):
            k_new = k - (np.log(k) - digamma(k) - s) / (1 / k - polygamma(1, k))
            if abs(k_new - k) < 1e-14 * k:
                k = k_new
                break
            k = k_new
        return float(k)

    d1 = gamma_shape_mle(A)
    d2 = gamma_shape_mle(B)
    return dict(delta1=d1, delta2=d2, alpha=len(A) * d1, beta=float(A.sum()), zeta=len(B) * d2,
                gamma=float(B.sum()), eta=1.0, sigma=1.0, u=1.0, v=1.0, repulsion=True, maxK=0)


def point_scores(D, logD, clusts, sizes, P, r, p, i):
    """Candidate labels and logprobs (mcmc.jl:247) for 0-based point i, which is removed first
    (mcmc.jl:193-194).  Returns (cands (1-based labels), logprobs)."""
    n = len(clusts)
    clusts = clusts.copy()
    sizes = sizes.copy()
    sizes[clusts[i] - 1] -= 1
    clusts[i] = -1
    d1, d2, al, be, ze, ga = (P[k] for k in ("delta1", "delta2", "alpha", "beta", "zeta", "gamma"))
    abratio = al * np.log(be) - gammaln(al)
    zgratio = ze * np.log(ga) - gammaln(ze)
    C_i = np.flatnonzero(sizes > 0) + 1
    K_i = len(C_i)
    if (P["maxK"] == 0 or K_i < P["maxK"]) and K_i < n:
        cands = np.concatenate([C_i, [np.flatnonzero(sizes == 0)[0] + 1]])
    else:
        cands = C_i
    m = len(cands)
    sumD = np.zeros(n)
    sumL = np.zeros(n)
    for k in C_i:
        mem = np.flatnonzero(clusts == k)
        # sequential ascending-member sums (np.sum would pairwise-reassociate)
        s = 0.0
        t = 0.0
        for j in mem:
            s += D[i, j]
            t += logD[i, j]
        sumD[k - 1] = s
        sumL[k - 1] = t
    L1 = np.zeros(m)
    lpr = np.zeros(m)
    for k in range(m):
        c = cands[k] - 1
        sz = sizes[c]
        if sz == 0:
            lpr[k] = np.log(K_i + 1) + r * np.log(1 - p)
            L1[k] = 0.0
        else:
            a_i = al + d1 * sz
            b_i = be + sumD[c]
            L1[k] = gammaln(a_i) + abratio - a_i * np.log(b_i) + (d1 - 1) * sumL[c] - sz * gammaln(d1)
            lpr[k] = np.log(sz + 1) + np.log(p) + np.log(sz - 1 + r) - np.log(sz)
    L2p = {}
    L2_i = 0.0
    for k in C_i:
        c = k - 1
        z_i = ze + d2 * sizes[c]
        g_i = ga + sumD[c]
        L2p[k] = gammaln(z_i) - z_i * np.log(g_i) + zgratio + (d2 - 1) * sumL[c] - sizes[c] * gammaln(d2)
    for k in C_i:
        L2_i += L2p[k]
    L2 = np.array([L2_i - L2p[c] if sizes[c - 1] != 0 else L2_i for c in cands])
    rep = 1.0 if P["repulsion"] else 0.0
    logprobs = lpr + (L1 + (L2 * rep if rep else 0.0))
    return cands, logprobs


def sample_logweights(logprobs, seed, sweep, i, keys):
    """keys[k]: the uniform of candidate k is u(seed, sweep, i, keys[k]) — its cluster label, 0 for the new cluster."""
    lp = logprobs - logprobs.min()
    g = np.array([-np.log(-np.log(uniform(seed, sweep, i, int(keys[k])))) for k in range(len(lp))])
    return int(np.argmax(g + lp))  # first maximum


def sweep(D, logD, clusts, sizes, P, r, p, seed, sweep_index):
    """One sample_labels_Gibbs! pass; clusts/sizes are modified in place; returns K."""
    n = len(clusts)
    for i in range(n):
        cands, lp = point_scores(D, logD, clusts, sizes, P, r, p, i)
        removed = sizes.copy()
        removed[clusts[i] - 1] -= 1
        keys = [c if removed[c - 1] > 0 else 0 for c in cands]   # the new-cluster candidate (last) has key 0
        k = sample_logweights(lp, seed, sweep_index, i, keys)
        sizes[clusts[i] - 1] -= 1
        clusts[i] = cands[k]
        sizes[cands[k] - 1] += 1
    return int(np.sum(sizes > 0))


def loglik(D, logD, clusts, sizes, P):
    d1, d2, al, be, ze, ga = (P[k] for k in ("delta1", "delta2", "alpha", "beta", "zeta", "gamma"))
    abratio = al * np.log(be) - gammaln(al)
    zgratio = ze * np.log(ga) - gammaln(ze)
    C = np.flatnonzero(sizes > 0) + 1
    members = [np.flatnonzero(clusts == k) for k in C]
    L1 = 0.0
    for k, mem in zip(C, members):
        sz = int(sizes[k - 1])
        pairs = sz * (sz - 1) // 2
        a = al + d1 * pairs
        b = be + D[np.ix_(mem, mem)].sum() / 2
        L1 += (d1 - 1) * logD[np.ix_(mem, mem)].sum() / 2 - pairs * gammaln(d1) + abratio + gammaln(a) - a * np.log(b)
    L2 = 0.0
    for x in range(len(C)):
        for y in range(x + 1, len(C)):
            pairs = int(sizes[C[x] - 1]) * int(sizes[C[y] - 1])
            z = ze + d2 * pairs
            g = ga + D[np.ix_(members[x], members[y])].sum()
            L2 += (d2 - 1) * logD[np.ix_(members[x], members[y])].sum() - pairs * gammaln(d2) + zgratio + gammaln(z) - z * np.log(g)
    return float(L1 + (L2 if P["repulsion"] else 0.0))


def logprior(sizes, r, p, P):
    from scipy.stats import beta as beta_dist, gamma as gamma_dist
    nz = sizes[sizes > 0].astype(float)
    K = len(nz)
    n = int(sizes.sum())
    L = (gammaln(K + 1) + (n - K) * np.log(p) + (r * K) * np.log(1 - p) - K * gammaln(r)
         + gamma_dist.logpdf(r, a=P["eta"], scale=1 / P["sigma"]) + beta_dist.logpdf(p, P["u"], P["v"]))
    L += np.sum(np.log(nz) + gammaln(nz + r - 1))
    return float(L)


def sortlabels(x):
    seen = {}
    out = np.empty_like(x)
    for t, v in enumerate(x):
        if v not in seen:
            seen[v] = len(seen) + 1
        out[t] = seen[v]
    return out


def adjacencymatrix(x):
    return x[:, None] == x[None, :]


# ------------------------------------------------------------------------------------------------------
# Split–merge step, as written: sample_labels! (mcmc.jl:356-479), sample_labels_Gibbs_restricted! (:259-354)
# ------------------------------------------------------------------------------------------------------
def uniform_mh(seed: int, it: int, mh: int, draw: int) -> float:
    c = philox4x32_10((draw, mh, it & M32, (it >> 32) & M32), (seed & M32, ((seed >> 32) & M32) ^ 0x4D485F52))
    bits = ((c[0] << 32) | c[1]) >> 12
    return (bits + 0.5) * 2.0 ** -52


def _seqsum(M, x, mem):
    s = 0.0
    for y in mem:
        s += M[x, y]
    return s


def restricted_scan(D, logD, clusts, sizes, P, r, p, items, cand, final_clusts, seed, it, mh, scan):
    d1, d2, al, be, ze, ga = (P[k] for k in ("delta1", "delta2", "alpha", "beta", "zeta", "gamma"))
    abratio = al * np.log(be) - gammaln(al)
    zgratio = ze * np.log(ga) - gammaln(ze)
    C = np.flatnonzero(sizes > 0) + 1                       # frozen at entry (:273)
    cinds = [int(np.flatnonzero(C == c)[0]) for c in cand]  # indexin(candidate_clusts, C)
    m = len(items)
    ltp = 0.0
    with np.errstate(all="ignore"):
        for q, x in enumerate(items):
            sizes[clusts[x] - 1] -= 1
            clusts[x] = -1
            a_i = np.zeros(2); b_i = np.zeros(2)
            for k in range(2):
                mem = np.flatnonzero(clusts == cand[k])
                a_i[k] = al + d1 * sizes[cand[k] - 1]
                b_i[k] = be + _seqsum(D, x, mem)
            need = sorted(set([0, 1] + cinds))              # only these entries of the K-vectors are ever read
            z_i = {}; g_i = {}; sl = {}
            for t in need:
                mem = np.flatnonzero(clusts == C[t])
                z_i[t] = ze + d2 * sizes[C[t] - 1]
                g_i[t] = ga + _seqsum(D, x, mem)
                sl[t] = _seqsum(logD, x, mem)
            L1 = np.zeros(2); lpr = np.zeros(2)
            for k in range(2):
                sz = sizes[cand[k] - 1]
                L1[k] = gammaln(a_i[k]) + abratio - a_i[k] * np.log(b_i[k]) + (d1 - 1) * sl[cinds[k]] - sz * gammaln(d1)
                lpr[k] = np.log(sz + 1) + np.log(p) + np.log(sz - 1 + r) - np.log(sz)
            L2p = {t: gammaln(z_i[t]) - z_i[t] * np.log(g_i[t]) + zgratio + (d2 - 1) * sl[t] - sizes[C[t] - 1] * gammaln(d2)
                   for t in need}
            L2_i = L2p[0] + L2p[1]                          # Q3
            L2 = np.array([L2_i - L2p[cinds[0]], L2_i - L2p[cinds[1]]])
            logprobs = lpr + (L1 + (L2 if P["repulsion"] else 0.0))
            if final_clusts is None:
                logprobs = logprobs - logprobs.min()        # sample_logweights mutates its argument
                base = 4 + m + 2 * m * scan + 2 * q
                g = np.array([-np.log(-np.log(uniform_mh(seed, it, mh, base + k))) for k in range(2)]) + logprobs
                k = int(np.argmax(g))
            else:
                k = 0 if final_clusts[x] == cand[0] else 1
            clusts[x] = cand[k]
            sizes[cand[k] - 1] += 1
            logprobs = logprobs + logprobs.min()            # Q2 (np.min propagates NaN like Julia's minimum)
            probs = np.exp(logprobs)
            probs = probs / (probs[0] + probs[1])
            ltp += np.log(probs[k])
    return float(ltp)


def mh_proposal(D, logD, clusts, sizes, K, P, r, p, numGibbs, seed, it, mh):
    """One proposal (mcmc.jl:374-473).  Returns (accept, split, skipped, final (clusts, sizes, K) or None, info)."""
    n = len(clusts)
    i = min(int(np.floor(uniform_mh(seed, it, mh, 0) * n)), n - 1)
    j = min(int(np.floor(uniform_mh(seed, it, mh, 1) * (n - 1))), n - 2)
    if j >= i:
        j += 1
    ci, cj = int(clusts[i]), int(clusts[j])
    if P["maxK"] > 0 and ci == cj and int(np.sum(sizes > 0)) >= P["maxK"]:
        return False, False, True, None, {}
    S = [k for k in np.flatnonzero((clusts == ci) | (clusts == cj)) if k != i and k != j]
    claunch, szlaunch, Klaunch = clusts.copy(), sizes.copy(), K
    if ci == cj:
        new = int(np.flatnonzero(sizes == 0)[0]) + 1
        claunch[i] = new
        szlaunch[ci - 1] -= 1
        szlaunch[new - 1] += 1
        Klaunch = K + 1
    cand = (int(claunch[i]), int(claunch[j]))
    for q, k in enumerate(S):
        claunch[k] = cand[0 if uniform_mh(seed, it, mh, 4 + q) < 0.5 else 1]
        szlaunch[clusts[k] - 1] -= 1
        szlaunch[claunch[k] - 1] += 1
    for s in range(numGibbs):
        restricted_scan(D, logD, claunch, szlaunch, P, r, p, S, cand, None, seed, it, mh, s)
    lg = gammaln
    if ci == cj:
        split = True
        ltp = restricted_scan(D, logD, claunch, szlaunch, P, r, p, S, cand, None, seed, it, mh, numGibbs)
        cfinal, szfinal, Kfinal = claunch, szlaunch, Klaunch
        lpr = (np.log(K + 1) + r * np.log(1 - p) - np.log(p) - lg(r) + lg(szfinal[cfinal[i] - 1] - 1 + r)
               + lg(szfinal[cfinal[j] - 1] - 1 + r) + np.log(szfinal[cfinal[i] - 1]) + np.log(szfinal[cfinal[j] - 1])
               - (lg(sizes[ci - 1] - 1 + r) + np.log(sizes[ci - 1])))
        lprop = ltp
    else:
        split = False
        cfinal, szfinal, Kfinal = claunch.copy(), szlaunch.copy(), Klaunch
        mem = np.flatnonzero(cfinal == ci)
        cfinal[mem] = cj
        szfinal[ci - 1] = 0
        szfinal[cj - 1] += len(mem)
        Kfinal -= 1
        lpr = (-(np.log(K) + r * np.log(1 - p) - np.log(p) - lg(r)) + lg(szfinal[cj - 1] - 1 + r) + np.log(szfinal[cj - 1])
               - (lg(sizes[ci - 1] - 1 + r) + lg(sizes[cj - 1] - 1 + r) + np.log(sizes[ci - 1]) + np.log(sizes[cj - 1])))
        ltp = restricted_scan(D, logD, claunch, szlaunch, P, r, p, S, cand, clusts, seed, it, mh, numGibbs)
        lprop = -ltp
    llr = loglik(D, logD, cfinal, szfinal, P) - loglik(D, logD, clusts, sizes, P)
    x = lpr + llr - lprop
    lar = np.nan if np.isnan(x) else min(0.0, x)
    lu = np.log(uniform_mh(seed, it, mh, 2))
    accept = bool(lu < lar)
    info = dict(i=i, j=j, nS=len(S), log_prior_ratio=float(lpr), log_lik_ratio=float(llr), log_proposal_ratio=float(lprop))
    return accept, split, False, (cfinal, szfinal, Kfinal), info


def sample_labels(D, logD, clusts, sizes, K, P, r, p, numMH, numGibbs, seed, it):
    """sample_labels! as written (Q1): returns (accept flags, split flags, K of the caller's state afterwards);
    the caller's clusts/sizes are swept in place only if no proposal was accepted."""
    state = (clusts, sizes, K)
    accept = [False] * numMH
    split = [False] * numMH
    for mh in range(numMH):
        a, s, skipped, final, _ = mh_proposal(D, logD, state[0], state[1], state[2], P, r, p, numGibbs, seed, it, mh)
        split[mh] = s
        if a:
            accept[mh] = True
            state = final                                   # rebinding: the caller's arrays are not touched
    Knew = sweep(D, logD, state[0], state[1], P, r, p, seed, it)
    return accept, split, (Knew if state[0] is clusts else K)


# ---------------------------------------------------------------------------------------------------------------
# Point estimation / clustering comparison (src/pointestimate.jl, src/summaries.jl:12-23).  Clustering.jl's
# randindex / varinfo / mutualinfo are third party (not under the reference checkout): written here from their
# textbook definitions on the contingency table, deliberately NOT through the four-sums route the C oracle takes.
# ---------------------------------------------------------------------------------------------------------------
def contingency(a, b):
    ua, ia = np.unique(a, return_inverse=True)
    ub, ib = np.unique(b, return_inverse=True)
    C = np.zeros((len(ua), len(ub)), dtype=np.int64)
    np.add.at(C, (ia, ib), 1)
    return C


def randindex(a, b):
    """(ARI, RI, Mirkin, Hubert) — Hubert & Arabie (1985) as Clustering.jl's randindex returns them."""
    C = contingency(a, b).astype(np.float64)
    n = C.sum()
    nis = (C.sum(axis=1) ** 2).sum()
    njs = (C.sum(axis=0) ** 2).sum()
    t1 = n * (n - 1) / 2
    t2 = (C ** 2).sum()
    t3 = 0.5 * (nis + njs)
    nc = (n * (n ** 2 + 1) - (n + 1) * nis - (n + 1) * njs + 2 * (nis * njs) / n) / (2 * (n - 1))
    A = t1 + t2 - t3
    Dd = -t2 + t3
    ari = 0.0 if t1 == nc else (A - nc) / (t1 - nc)
    return ari, A / t1, Dd / t1, (A - Dd) / t1


def entropy_of_labels(a):
    p = np.unique(a, return_counts=True)[1] / len(a)
    return float(-(p * np.log(p)).sum())


def mutualinfo(a, b, normed=True):
    C = contingency(a, b)
    n = C.sum()
    P = C / n
    pa, pb = P.sum(axis=1, keepdims=True), P.sum(axis=0, keepdims=True)
    nz = P > 0
    mi = float((P[nz] * np.log(P[nz] / (pa @ pb)[nz])).sum())
    if normed:
        return 2 * mi / (entropy_of_labels(a) + entropy_of_labels(b))
    return mi


def varinfo(a, b):
    return entropy_of_labels(a) + entropy_of_labels(b) - 2 * mutualinfo(a, b, normed=False)


def binderloss(a, b, normalised=True):
    """pointestimate.jl:68-76"""
    n = len(a)
    return randindex(a, b)[2] * (1 if normalised else n * (n - 1) // 2)


def infodist(a, b, normalised=True):
    """pointestimate.jl:89-99"""
    hu, hv = entropy_of_labels(a), entropy_of_labels(b)
    mi = mutualinfo(a, b, normed=False)
    return 1 - mi / max(hu, hv) if normalised else max(hu, hv) - mi


def evaluateclustering(clusts, truth):
    """summaries.jl:12-23"""
    n = len(clusts)
    ari, _, nbloss, _ = randindex(clusts, truth)
    vi = varinfo(clusts, truth)
    idd = infodist(clusts, truth, normalised=False)
    return dict(nbloss=nbloss, ari=ari, vi=vi, nvi=vi / np.log(n), id=idd, nid=idd / np.log(n),
                nmi=mutualinfo(clusts, truth))


def getpointestimate_mpel(samples, loss):
    """pointestimate.jl:36-58: upper-triangle loss matrix, symmetrised, column sums, first argmin (0-based)."""
    fn = {"binder": lambda x, y: randindex(x, y)[2], "omARI": lambda x, y: 1 - randindex(x, y)[0],
          "VI": varinfo, "ID": lambda x, y: infodist(x, y, normalised=False)}[loss]
    m = len(samples)
    L = np.zeros((m, m))
    for i in range(m):
        for j in range(i + 1, m):
            L[i, j] = fn(samples[i], samples[j])
    L = L + L.T
    cs = L.sum(axis=0)
    return int(np.argmin(cs)), L, cs


# ---------------------------------------------------------------------------------------------------------------
# Scalar updates: sample_r (mcmc.jl:94-136), sample_p (mcmc.jl:147-155) on the build's scalar stream
# (Philox key (seed_lo, seed_hi ^ 0x52505F5F), counter (draw, kind, iter_lo, iter_hi); Box–Muller normals,
# Marsaglia–Tsang gammas) and the host loop of runsampler (mcmc.jl:533-556).
# ---------------------------------------------------------------------------------------------------------------
class ScalarStream:
    def __init__(self, seed, it, kind):
        self.seed, self.it, self.kind, self.draw = seed, it, kind, 0

    def uniform(self):
        out = philox4x32_10([self.draw & 0xFFFFFFFF, self.kind, self.it & 0xFFFFFFFF, (self.it >> 32) & 0xFFFFFFFF],
                            [self.seed & 0xFFFFFFFF, ((self.seed >> 32) & 0xFFFFFFFF) ^ 0x52505F5F])
        self.draw += 1
        bits = ((int(out[0]) << 32) | int(out[1])) >> 12
        return (bits + 0.5) * 2.0 ** -52

    def normal(self):
        u1, u2 = self.uniform(), self.uniform()
        return math.sqrt(-2.0 * math.log(u1)) * math.cos(2 * math.pi * u2)

    def gamma(self, a):
        boost = 1.0
        if a < 1.0:
            boost = self.uniform() ** (1.0 / a)
            a += 1.0
        d = a - 1.0 / 3.0
        c = 1.0 / math.sqrt(9.0 * d)
        while True:
            z = self.normal()
            t = 1.0 + c * z
            u = self.uniform()
            if t <= 0:
                continue
            v = t * t * t
            if math.log(u) < 0.5 * z * z + d - d * v + d * math.log(v):
                return d * v * boost


def _logpdf_truncnorm0(x, mu, sd):
    from scipy.special import log_ndtr
    z = (x - mu) / sd
    return -0.5 * z * z - math.log(sd) - 0.5 * math.log(2 * math.pi) - float(log_ndtr(mu / sd))


def sample_r(seed, it, r, p, C, K, eta, sigma, proposalsd_r):
    s = ScalarStream(seed, it, 0)
    while True:
        rc = r + proposalsd_r * s.normal()
        if rc >= 0:
            break
    C = np.asarray(C, dtype=np.float64)
    lpc = (eta - 1) * math.log(rc) + K * (rc * math.log(1 - p) - float(gammaln(rc))) - rc * sigma
    lpo = (eta - 1) * math.log(r) + K * (r * math.log(1 - p) - float(gammaln(r))) - r * sigma
    for nk in C:
        lpc += float(gammaln(nk - 1 + rc))
        lpo += float(gammaln(nk - 1 + r))
    lpr = _logpdf_truncnorm0(rc, r, proposalsd_r) - _logpdf_truncnorm0(r, rc, proposalsd_r)
    if math.log(s.uniform()) < min(0.0, lpc - lpo - lpr):
        return rc, True
    return r, False


def sample_p(seed, it, K, n, r, u, v):
    s = ScalarStream(seed, it, 1)
    x = s.gamma(n - K + u)
    y = s.gamma(r * K + v)
    return x / (x + y)


def run_chain(D, clusts, P, r, p, numiters, burnin, thin, numGibbs, numMH, seed, eta=1.0, sigma=1.0, proposalsd_r=1.0,
              u=1.0, v=1.0):
    """runsampler's loop (mcmc.jl:533-556) as written: sample_r!, sample_p!, sample_labels!, record."""
    n = D.shape[0]
    logD = make_logD(D)
    clusts = clusts.copy()
    sizes, K = state_from_labels(clusts)
    rec = dict(r=[], p=[], K=[], loglik=[], logposterior=[], clusts=[], r_acc=[], sm_acc=[], sm_split=[], r_all=[], p_all=[])
    for i in range(1, numiters + 1):
        C = sizes[sizes > 0]
        r, acc = sample_r(seed, i - 1, r, p, C, K, eta, sigma, proposalsd_r)
        rec["r_acc"].append(acc)
        p = sample_p(seed, i - 1, K, n, r, u, v)
        rec["r_all"].append(r); rec["p_all"].append(p)
        if numMH > 0:
            a, s, K = sample_labels(D, logD, clusts, sizes, K, P, r, p, numMH, numGibbs, seed, i - 1)
            rec["sm_acc"] += list(np.atleast_1d(a)); rec["sm_split"] += list(np.atleast_1d(s))
        else:
            K = sweep(D, logD, clusts, sizes, P, r, p, seed, i - 1)
        if i > burnin and (i - burnin) % thin == 0:
            ll = loglik(D, logD, clusts, sizes, P)
            rec["clusts"].append(sortlabels(clusts)); rec["K"].append(K); rec["r"].append(r); rec["p"].append(p)
            rec["loglik"].append(ll); rec["logposterior"].append(ll + logprior(sizes, r, p, dict(P, eta=eta, sigma=sigma, u=u, v=v)))
    return rec
