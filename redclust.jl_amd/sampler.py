"""runsampler — host loop of the reference's MCMC (src/mcmc.jl:501-590) around the HIP sweep.

Kept on the host, as in the reference: the scalar r / p updates (mcmc.jl:80-155), the recording rule
(mcmc.jl:546-553) and the diagnostics (mcmc.jl:564-587).  On the device behind the C ABI: the Gibbs sweep,
loglik, logprior's size terms, label canonicalisation and the co-clustering accumulation.

The split–merge step (numMH > 0, src/mcmc.jl:356-474) runs through rc_splitmerge: scalar scans on the host as
in the reference, both log-likelihoods on the device.  `splitmerge="as_written"` reproduces the reference
literally, including its rebinding quirk (after an accepted proposal the iteration leaves
the caller's labels untouched); `splitmerge="intended"` keeps accepted proposals and sweeps them.

Not in this build: fitprior (params must be given) and the k-medoids initialisation (init must be
given)."""
from __future__ import annotations

import math
import time

import numpy as np
from scipy.special import gammaln, log_ndtr

from ._lib import Context
from .types import MCMCData, MCMCOptionsList, MCMCResult, MCMCState, PriorHyperparamsList


def _logpdf_truncnorm_lower0(x, mu, sd):
    """logpdf of truncated(Normal(mu, sd), lower=0, upper=Inf) at x ≥ 0  (Distributions.truncated)."""
    z = (x - mu) / sd
    return -0.5 * z * z - math.log(sd) - 0.5 * math.log(2 * math.pi) - log_ndtr(mu / sd)


def _rand_truncnorm_lower0(rng, mu, sd):
    while True:  # simple rejection; acceptance ≥ 1/2 whenever mu ≥ 0 (r is always positive)
        x = rng.normal(mu, sd)
        if x >= 0:
            return x


def sample_r(rng, r, p, C, K, eta, sigma, proposalsd_r):
    """src/mcmc.jl:94-136: MH step for r with a Normal proposal truncated to [0, ∞)."""
    r_candidate = _rand_truncnorm_lower0(rng, r, proposalsd_r)
    C = np.asarray(C, dtype=np.float64)
    lpc = (eta - 1) * math.log(r_candidate) + K * (r_candidate * math.log(1 - p) - gammaln(r_candidate)) - r_candidate * sigma
    lpo = (eta - 1) * math.log(r) + K * (r * math.log(1 - p) - gammaln(r)) - r * sigma
    lpc += float(np.sum(gammaln(C - 1 + r_candidate)))
    lpo += float(np.sum(gammaln(C - 1 + r)))
    logproposalratio = _logpdf_truncnorm_lower0(r_candidate, r, proposalsd_r) - _logpdf_truncnorm_lower0(r, r_candidate, proposalsd_r)
    if math.log(rng.uniform()) < min(0.0, lpc - lpo - logproposalratio):
        return r_candidate, True
    return r, False


def sample_p(rng, K, n, r, u, v):
    """src/mcmc.jl:147-155: rand(Beta(n - K + u, r K + v))."""
    return float(rng.beta(n - K + u, r * K + v))


def iac_ess_acf(x):
    """src/utils.jl:41-46 with StatsBase.autocor's default lags 0:min(n-1, round(10·log10 n))."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    if n < 2:
        return float("nan"), float("nan"), np.ones(min(n, 1))
    maxlag = min(n - 1, int(round(10 * math.log10(n))))
    z = x - x.mean()
    den = float(np.dot(z, z))
    acf = np.array([float(np.dot(z[: n - k], z[k:])) / den if den > 0 else float("nan") for k in range(maxlag + 1)])
    iac = float(np.sum(acf) * 2)
    return iac, n / iac, acf


def _mean_and_var(x):
    x = np.asarray(x, dtype=np.float64)
    return float(x.mean()), float(x.var(ddof=1)) if len(x) > 1 else float("nan")


def runsampler(data: MCMCData, options: MCMCOptionsList | None = None, params: PriorHyperparamsList | None = None,
               init: MCMCState | None = None, *, verbose: bool = True, seed: int = 0, rng=None, device: int = 0,
               kcap: int = 0, ctx: Context | None = None, rp_trace=None, splitmerge: str = "as_written",
               host_logD=None, engine: str | None = None, mode: str = "incremental") -> MCMCResult:
    """runsampler(data, options, params, init; verbose) — src/mcmc.jl:501-590.

    seed keys the counter-based streams of the label draws, the split–merge proposals and the scalar r / p updates
    (DESIGN.md).  engine="native" (default) runs the whole iteration loop inside the library (rc_run_chain);
    engine="python" keeps the loop in this function and draws r / p from `rng` (a numpy Generator; giving `rng`
    selects it).  rp_trace=(r_seq, p_seq) teacher-forces r and p instead (parity tests).
    mode="incremental" (default) keeps the exact row-sum table up to date under label changes instead of recomputing it
    from the matrices in every sweep — bit-identical results (integer sums), no matrix traffic while labels are stable;
    mode="full" recomputes per sweep as the reference re-reads D and logD (the data flow bench.py measures)."""
    if mode not in ("incremental", "full"):
        raise ValueError("mode must be 'incremental' or 'full'")
    if engine is None:
        engine = "python" if rng is not None else "native"
    if engine not in ("native", "python"):
        raise ValueError("engine must be 'native' or 'python'")
    options = options or MCMCOptionsList()
    if params is None:
        raise NotImplementedError("fitprior is outside this build's scope (SURVEY.md §8): pass params explicitly")
    if init is None:
        raise NotImplementedError("k-medoids initialisation is outside this build's scope (SURVEY.md §8): pass init")
    if splitmerge not in ("as_written", "intended"):
        raise ValueError("splitmerge must be 'as_written' or 'intended'")
    out = print if verbose else (lambda *a, **k: None)
    rng = rng or np.random.default_rng(seed)
    n = data.n
    numiters, burnin, thin, numsamples = options.numiters, options.burnin, options.thin, options.numsamples
    own_ctx = ctx is None
    if own_ctx:
        ctx = (Context.from_points(data.points, device=device, kcap=kcap) if data.points is not None
               else Context(data.D, device=device, kcap=kcap))
    try:
        ctx.set_params(**params.as_dict())
        ctx.set_state(init.clusts)
        ctx.set_mode(mode)
        ctx.cocluster_reset()
        numMH = options.numMH
        if numMH > 0:
            if data.points is not None and host_logD is None:
                # MCMCData(points): the split–merge scans must see the SAME D and logD as the device's log-likelihoods —
                # the matrix the device computed from the points, not a second evaluation by numpy
                ctx.attach_host_matrices(ctx.get_matrix(0), ctx.get_matrix(1))
            else:
                ctx.attach_host_matrices(data.D, data.logD if host_logD is None else host_logD)
        result = MCMCResult.allocate(data, options, params)
        state = MCMCState(init.clusts, init.r, init.p)
        out("Run MCMC")
        out(f"Setup: {numiters} iterations, {numsamples} samples, {n} observations.")
        j = 0
        t0 = time.perf_counter()
        if engine == "native":
            ch = ctx.run_chain(numiters, burnin, thin, options.numGibbs, numMH, seed, state.r, state.p,
                               params.proposalsd_r, splitmerge=splitmerge, rp_trace=rp_trace)
            for j in range(ch["num_samples"]):
                result.clusts[j][:] = ch["clusts"][j]
            for k in ("K", "r", "p", "loglik", "logposterior"):
                getattr(result, k)[:] = ch[k]
            result.r_acceptances[:] = ch["r_acceptances"]
            result.splitmerge_acceptances[:] = ch["splitmerge_acceptances"]
            result.splitmerge_splits[:] = ch["splitmerge_splits"]
            state.r, state.p = ch["r_final"], ch["p_final"]
            numiters_py = 0
        else:
            numiters_py = numiters
        for i in range(1, numiters_py + 1):
            if rp_trace is None:
                C = state.clustsizes[state.clustsizes > 0]
                state.r, acc = sample_r(rng, state.r, state.p, C, state.K, params.eta, params.sigma, params.proposalsd_r)
                result.r_acceptances[i - 1] = acc                                   # mcmc.jl:538
                state.p = sample_p(rng, state.K, n, state.r, params.u, params.v)    # mcmc.jl:539
            else:
                state.r, state.p = float(rp_trace[0][i - 1]), float(rp_trace[1][i - 1])
            accepted_any = False
            if numMH > 0:                                                          # sample_labels!, mcmc.jl:372-474
                if splitmerge == "as_written":
                    ctx.checkpoint()
                for mh in range(numMH):
                    acc, spl = ctx.splitmerge(state.r, state.p, options.numGibbs, seed, i - 1, mh)
                    result.splitmerge_acceptances[(i - 1) * numMH + mh] = acc      # mcmc.jl:541-543
                    result.splitmerge_splits[(i - 1) * numMH + mh] = spl
                    accepted_any |= acc
            if accepted_any and splitmerge == "as_written":
                # Q1: `state = finalstate` (mcmc.jl:470) rebound a local name; the accepted proposal and the closing
                # Gibbs scan (mcmc.jl:477) acted on an object the caller never sees
                ctx.restore()
            else:
                ctx.gibbs_sweep(state.r, state.p, seed, i - 1)                     # mcmc.jl:540 → :477
            record = i > burnin and (i - burnin) % thin == 0                       # mcmc.jl:546
            if record or rp_trace is None:
                # sample_r!/sample_p! of the next iteration need K and the cluster sizes (mcmc.jl:84-89,139): read
                # from the host-mapped sweep summary, no device copy; labels are pulled only with a recorded sample
                _, state.clustsizes, state.K = ctx.get_state(want_labels=False)
            if record:
                result.clusts[j][:] = ctx.record_sample(True)                      # sortlabels, mcmc.jl:547
                result.K[j], result.r[j], result.p[j] = state.K, state.r, state.p  # mcmc.jl:548-550
                result.loglik[j] = ctx.loglik()                                    # mcmc.jl:551
                result.logposterior[j] = result.loglik[j] + ctx.logprior(state.r, state.p)  # mcmc.jl:552
                j += 1
        runtime = time.perf_counter() - t0
        out("Computing summary statistics and diagnostics.")
        result.posterior_coclustering = ctx.cocluster(max(numsamples, 1)) if numsamples > 0 else np.zeros((n, n))
        result.K_iac, result.K_ess, result.K_acf = iac_ess_acf(result.K)           # mcmc.jl:564-573
        result.K_mean, result.K_variance = _mean_and_var(result.K)
        result.r_iac, result.r_ess, result.r_acf = iac_ess_acf(result.r)
        result.r_mean, result.r_variance = _mean_and_var(result.r)
        result.p_iac, result.p_ess, result.p_acf = iac_ess_acf(result.p)
        result.p_mean, result.p_variance = _mean_and_var(result.p)
        result.splitmerge_acceptance_rate = float(np.mean(result.splitmerge_acceptances)) if numMH > 0 else 0.0  # :576-580
        result.r_acceptance_rate = float(np.mean(result.r_acceptances))
        result.runtime = runtime
        result.mean_iter_time = runtime / numiters
        state.clusts, state.clustsizes, state.K = ctx.get_state()
        result.final_state = state
        return result
    finally:
        if own_ctx:
            ctx.close()
