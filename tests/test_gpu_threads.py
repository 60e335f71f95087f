"""Several chains driven from several host threads of ONE process (-m gpu) — the shape of rc_run_chains (one host thread and
one context per GPU), exercised on the one GPU there is: every thread owns a context on device 0 and runs
rc_run_chain concurrently with the others.  The chains share nothing but the device (the resolver launches of different
contexts on one device are chained, DESIGN.md §3 "Co-residency across contexts") and the process-wide pieces of the library
(per-device resolver lock, running-chain count, RCCL loader, thread-local error strings), so every chain must equal the same
chain run alone, bit for bit.  tools/chaos_threads.sh repeats this file on the chaos build (block-dependent random delays inside
the resolver's rounds)."""
import threading

import numpy as np
import pytest

import redclust_amd as rc

pytestmark = pytest.mark.gpu

FIELDS = ("clusts", "K", "r", "p", "loglik", "logposterior", "r_acceptances", "splitmerge_acceptances", "splitmerge_splits",
          "r_all", "p_all", "final", "cocluster", "r_final", "p_final", "num_samples")


def _chain(D, L, P, init, seed, numMH, iters, workers, out, key, barrier=None):
    try:
        ctx = rc.Context(D, device=0)
        ctx.set_params(**P)
        ctx.set_state(init)
        ctx.cocluster_reset()
        ctx.set_option("chain_workers", workers)          # per context (rc_set_option): nothing reads the environment per chain
        ctx.set_option("chain_depth", 8)
        if numMH:
            ctx.attach_host_matrices(D, L)
        if barrier is not None:
            barrier.wait(timeout=600)
        ch = ctx.run_chain(iters, 10, 3, 5, numMH, seed, 1.0, 0.5, 1.0)
        ch["final"] = ctx.get_state()[0]
        ch["cocluster"] = ctx.cocluster(max(ch["num_samples"], 1))
        ctx.close()
        out[key] = ch
    except BaseException as e:   # noqa: BLE001  (reported by the main thread)
        out[key] = e
        if barrier is not None:
            barrier.abort()


@pytest.mark.parametrize("nthreads", [4, 8])
def test_concurrent_chains_on_one_device_equal_the_chains_run_alone(nthreads):
    n, K, iters = 2000, 20, 60
    data = rc.generatemixture(n, K, seed=31, sigma=0.25)       # overlapping clusters: labels move, births and deaths, several resolver rounds
    D, truth = data["distancematrix"], data["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    L = np.log(np.where(np.eye(n, dtype=bool), 1.0, D))
    init = truth.copy()
    idx = np.random.default_rng(5).choice(n, n // 20, replace=False)
    init[idx] = np.random.default_rng(6).integers(1, K + 1, size=len(idx))
    jobs = [(100 + q, q % 2) for q in range(nthreads)]         # distinct seeds; numMH alternates 0 / 1
    alone, together = {}, {}
    for seed, numMH in jobs:
        _chain(D, L, P, init, seed, numMH, iters, 2, alone, seed)
        assert not isinstance(alone[seed], BaseException), alone[seed]
    barrier = threading.Barrier(nthreads)
    th = [threading.Thread(target=_chain, args=(D, L, P, init, seed, numMH, iters, 2, together, seed, barrier)) for seed, numMH in jobs]
    for t in th: t.start()
    for t in th: t.join(timeout=900)
    assert not any(t.is_alive() for t in th), "a chain thread hangs"
    moved = 0
    for seed, numMH in jobs:
        a, b = alone[seed], together.get(seed)
        assert b is not None and not isinstance(b, BaseException), (seed, b)
        for f in FIELDS:
            assert np.array_equal(a[f], b[f]), (seed, numMH, f)
        moved += int(np.sum(a["clusts"][0] != a["clusts"][-1]))
    assert moved > 20 * nthreads                                # the chains move
    # distinct seeds gave distinct chains (the comparison above is not vacuous)
    assert not np.array_equal(alone[jobs[0][0]]["clusts"], alone[jobs[2][0]]["clusts"])


def test_concurrent_sweeps_and_observables_from_threads():
    """The plain ABI from several threads: every thread owns a context (different data sizes), sweeps asynchronously, records
    samples and reads the log-likelihood while the others do the same; results equal a serial run of the same calls."""
    specs = [(700, 6, 0.45, 11), (1500, 12, 0.3, 12), (2500, 16, 0.2, 13), (333, 5, 0.5, 14)]

    def job(spec, out, key, barrier=None):
        try:
            n, K, sigma, seed = spec
            data = rc.generatemixture(n, K, seed=seed, sigma=sigma, dim=max(K, 6))
            D, truth = data["distancematrix"], data["clusts"]
            P = rc.likelihood_hyperparams(D, truth)
            init = np.random.default_rng(seed).integers(1, 2 * K, n).astype(np.int64)
            ctx = rc.Context(D, device=0)
            ctx.set_params(**P); ctx.set_state(init); ctx.cocluster_reset()
            if barrier is not None:
                barrier.wait(timeout=600)
            lls = []
            for t in range(10):
                ctx.gibbs_sweep(1.0 + 0.1 * t, 0.5, seed, t, blocking=False)
                if t % 3 == 2:
                    ctx.record_sample(False)
                    lls.append(ctx.loglik())
            ctx.synchronize()
            out[key] = dict(state=ctx.get_state(), lls=np.array(lls), counts=ctx.cocluster_counts(), changes=ctx.sweep_stats()["n_changes"])
            ctx.close()
        except BaseException as e:   # noqa: BLE001
            out[key] = e
            if barrier is not None:
                barrier.abort()

    serial, conc = {}, {}
    for q, sp in enumerate(specs):
        job(sp, serial, q)
        assert not isinstance(serial[q], BaseException), serial[q]
    barrier = threading.Barrier(len(specs))
    th = [threading.Thread(target=job, args=(sp, conc, q, barrier)) for q, sp in enumerate(specs)]
    for t in th: t.start()
    for t in th: t.join(timeout=900)
    assert not any(t.is_alive() for t in th)
    for q in range(len(specs)):
        a, b = serial[q], conc.get(q)
        assert b is not None and not isinstance(b, BaseException), (q, b)
        assert np.array_equal(a["state"][0], b["state"][0]) and np.array_equal(a["state"][1], b["state"][1]) and a["state"][2] == b["state"][2], q
        assert np.array_equal(a["lls"], b["lls"]) and np.array_equal(a["counts"], b["counts"]) and a["changes"] == b["changes"], q
