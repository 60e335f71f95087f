"""Randomised differential check of the sweep against the oracle: random sizes, cluster counts, overlaps, capacities, batch
capacities, storage widths, stored / derived logD, maxK and repulsion, from random labels (births, deaths, relabelings).
usage: python3 tests/fuzz_parity.py [cases] [first_seed]   (test_gpu_fuzz.py runs a short batch in the suite)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import redclust_amd as rc
import oracle_lib as O
def run(cases, first, nlo=40, nhi=1500, kmax=25):
    saved = {k: os.environ.get(k) for k in ("RC_RES_MAXB", "RC_RES_ONE_STREAM", "RC_SCORE_CACHE")}
    bad = 0
    for seed in range(first, first + cases):
        g = np.random.default_rng(seed)
        # resolver score cache: off / filled in every sweep / adaptive (the default) — a separate generator keeps the cases of a seed what they were
        sc = ["0", "1", "1", None][int(np.random.default_rng(seed + 77777).integers(0, 4))]
        if sc is None: os.environ.pop("RC_SCORE_CACHE", None)
        else: os.environ["RC_SCORE_CACHE"] = sc
        n = int(g.integers(nlo, nhi)); K = int(g.integers(2, kmax)); dim = int(g.integers(max(2, K), K + 6))
        sigma = float(g.uniform(0.15, 0.9))
        data = rc.generatemixture(n, K, seed=seed, sigma=sigma, dim=dim)
        sh = g.permutation(n)
        D = np.ascontiguousarray(data["distancematrix"][np.ix_(sh, sh)]); truth = data["clusts"][sh]
        P = dict(rc.likelihood_hyperparams(D, truth), repulsion=bool(g.random() < 0.8), maxK=int(g.choice([0, 0, K + 3, 2 * K])))
        bits = int(g.choice([64, 64, 32])); stored = bool(g.random() < 0.4) or bits == 32
        os.environ["RC_RES_MAXB"] = str(int(g.choice([512, 512, 64, 16])))
        os.environ["RC_RES_ONE_STREAM"] = str(int(g.integers(0, 2)))
        kcap = int(g.choice([min(n, 4096), min(n, 4 * K + 64), 0, 8]))   # 0 / 8: the capacity grows on demand, in the middle of sweeps
        orc0 = O.Oracle(D, P)
        ctx = rc.Context(D, logD=orc0.logD if stored else None, kcap=kcap, storage_bits=bits)
        ctx.set_params(**P)
        L = ctx.get_matrix(1); Dd = ctx.get_matrix(0)
        eD, eL = ctx.debug_rowsums(1)[2:4] if False else (None, None)
        init = g.integers(1, int(g.integers(1, min(n, 3 * K) + 1)) + 1, n).astype(np.int64)
        if g.random() < 0.5:      # a few percent of the labels re-drawn around the generating partition: the symmetric kernels' regime
            init = truth.copy(); idx = g.choice(n, max(1, int(n * g.uniform(0.0, 0.05))), replace=False); init[idx] = g.integers(1, K + 1, len(idx))
        if P["maxK"]: init = (init - 1) % P["maxK"] + 1
        ctx.set_state(init)
        eD, eL = ctx.debug_rowsums(int(init[0]))[2:4]
        orc = O.Oracle(Dd, P, logD=L, eL=eL, eD=eD)
        orc.set_state(init)
        mode = "incremental" if g.random() < 0.5 else "full"
        ctx.set_mode(mode)
        if g.random() < 0.25:             # the path of blocks with many chunks (n > 128 x #CUs): pi[] / slot_of[] from global memory
            try: ctx.set_option("lds_point_cache", 0)
            except rc.RedClustHIPError: pass   # (an older build under comparison: RC_LIB_PATH)

        ok = True
        try:
            for t in range(5):
                r, p = float(g.uniform(0.3, 3.0)), float(g.uniform(0.05, 0.95))
                ctx.gibbs_sweep(r, p, seed, t, blocking=bool(t & 1))
                orc.sweep_stable(r, p, seed, t)
                lab, sizes, Kc = ctx.get_state()
                if not (np.array_equal(lab, orc.clusts) and np.array_equal(sizes, orc.sizes) and Kc == orc.K and ctx.sweep_stats()["n_changes"] == orc.last_changes):
                    ok = False
                    print(f"MISMATCH seed {seed} sweep {t}: n={n} K={K} sigma={sigma:.2f} bits={bits} stored={stored} kcap={kcap} maxK={P['maxK']} rep={P['repulsion']} mode={mode} env={os.environ['RC_RES_MAXB']},{os.environ['RC_RES_ONE_STREAM']},cache={os.environ.get('RC_SCORE_CACHE')} differing {int(np.sum(lab != orc.clusts))} stats {ctx.sweep_stats()} oracle changes {orc.last_changes}")
                    break
        except rc.RedClustHIPError as e:
            if "RC_ERR_CAPACITY" not in str(e):
                ok = False; print(f"ERROR seed {seed}: {e}")
        bad += not ok
        ctx.close()

    for k, v in saved.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    return bad


def run_wide(cases, first):
    """Wide contexts (more than 4096 clusters; k_sweep_wide, k_derive_wide): n between 4150 and 5200, most points clusters of their own,
    capacities that leave a wide context room to overflow again (explicit kcap between 4096 and n), both modes, both storage widths,
    three sweeps against the oracle.  Slow (hundreds of milliseconds per sweep): a handful of cases."""
    bad = 0
    for seed in range(first, first + cases):
        g = np.random.default_rng(seed)
        n = int(g.integers(4150, 5200)); K = int(g.integers(3, 40)); dim = int(g.integers(max(2, K), K + 6))
        sigma = float(g.uniform(0.1, 0.6))
        data = rc.generatemixture(n, K, seed=seed, sigma=sigma, dim=dim)
        sh = g.permutation(n)
        D = np.ascontiguousarray(data["distancematrix"][np.ix_(sh, sh)]); truth = data["clusts"][sh]
        P = dict(rc.likelihood_hyperparams(D, truth), repulsion=bool(g.random() < 0.5), maxK=0)
        bits = int(g.choice([64, 32])); stored = bool(g.random() < 0.5) or bits == 32
        kcap = int(g.choice([0, 4096, int(g.integers(4097, n)), n]))
        orc0 = O.Oracle(D, P)
        init = np.arange(1, n + 1, dtype=np.int64)
        merged = g.choice(n, int(g.integers(0, n - 4100)), replace=False)       # some points share a few clusters: K stays above 4096
        init[merged] = init[merged[: max(1, len(merged) // 50)]][g.integers(0, max(1, len(merged) // 50), len(merged))] if len(merged) else init[merged]
        ok = True
        try:
            ctx = rc.Context(D, logD=orc0.logD if stored else None, kcap=kcap, storage_bits=bits)
            ctx.set_params(**P)
            L = ctx.get_matrix(1); Dd = ctx.get_matrix(0)
            ctx.set_state(init)
            eD, eL = ctx.debug_rowsums(int(init[0]))[2:4]
            orc = O.Oracle(Dd, P, logD=L, eL=eL, eD=eD)
            orc.set_state(init)
            mode = "incremental" if g.random() < 0.5 else "full"
            ctx.set_mode(mode)
            for t in range(3):
                r, p = float(g.uniform(0.3, 3.0)), float(g.uniform(0.05, 0.95))
                ctx.gibbs_sweep(r, p, seed, t, blocking=bool(t & 1))
                orc.sweep_stable(r, p, seed, t)
                lab, sizes, Kc = ctx.get_state()
                if not (np.array_equal(lab, orc.clusts) and np.array_equal(sizes, orc.sizes) and Kc == orc.K and ctx.sweep_stats()["n_changes"] == orc.last_changes):
                    ok = False
                    print(f"MISMATCH (wide) seed {seed} sweep {t}: n={n} K0={len(np.unique(init))} bits={bits} stored={stored} kcap={kcap} rep={P['repulsion']} mode={mode} differing {int(np.sum(lab != orc.clusts))} stats {ctx.sweep_stats()} oracle K {orc.K} changes {orc.last_changes} capacity {ctx.capacity_info()}")
                    break
            if ok:      # the observables of the state reached: log-likelihood (block sums tiled over thousands of slots), recorded sample
                ll, ref = ctx.loglik(), orc.loglik_stable()
                canon = ctx.record_sample(True)
                ref_c = np.zeros(n, np.int64)
                O.lib().orc_sortlabels(n, orc.clusts, ref_c)
                if not (abs(ll - ref) <= 1e-9 * abs(ref) and np.array_equal(canon, ref_c)):
                    ok = False
                    print(f"MISMATCH (wide, observables) seed {seed}: n={n} bits={bits} stored={stored} kcap={kcap} mode={mode} loglik {ll} vs {ref} canonical labels equal {np.array_equal(canon, ref_c)} K {orc.K}")
            ctx.close()
        except rc.RedClustHIPError as e:
            ok = False; print(f"ERROR (wide) seed {seed}: {e}")
        bad += not ok
    return bad


def run_chains(cases, first):
    """rc_run_chain with split-merge proposals: the speculative pipeline at a random depth / worker count against the synchronous
    loop (RC_CHAIN_PIPELINE=0) on random small problems — every output array and the final state must be identical."""
    keys = ("RC_CHAIN_PIPELINE", "RC_CHAIN_DEPTH", "RC_CHAIN_WORKERS")
    saved = {k: os.environ.get(k) for k in keys}
    bad = 0
    for seed in range(first, first + cases):
        g = np.random.default_rng(seed)
        n = int(g.integers(30, 400)); K = int(g.integers(2, 12)); dim = int(g.integers(max(2, K), K + 4))
        data = rc.generatemixture(n, K, seed=seed, sigma=float(g.uniform(0.3, 0.9)), dim=dim)
        D, truth = data["distancematrix"], data["clusts"]
        P = dict(rc.likelihood_hyperparams(D, truth), maxK=int(g.choice([0, 0, 2 * K])))
        L = np.log(D + np.eye(n))
        init = g.integers(1, int(g.integers(1, 2 * K + 1)) + 1, n).astype(np.int64)
        if P["maxK"]: init = (init - 1) % P["maxK"] + 1
        iters, burn, thin = int(g.integers(5, 60)), int(g.integers(0, 5)), int(g.integers(1, 4))
        numGibbs, numMH = int(g.integers(0, 6)), int(g.integers(1, 4))
        mode = str(g.choice(["as_written", "intended"])); stored = bool(g.random() < 0.5)
        if np.random.default_rng(seed + 4242).random() < 0.2: numMH = 0      # the pipelined loop without proposals (a separate generator keeps the other cases of a seed what they were)
        outs = []
        for env in ({"RC_CHAIN_PIPELINE": "0"}, {"RC_CHAIN_DEPTH": str(int(g.integers(1, 30))), "RC_CHAIN_WORKERS": str(int(g.integers(1, 9)))}):
            for k in keys: os.environ.pop(k, None)
            os.environ.update(env)
            ctx = rc.Context(D, logD=L if stored else None, kcap=n); ctx.set_params(**P); ctx.set_state(init); ctx.cocluster_reset()
            ctx.attach_host_matrices(D, L)
            ch = ctx.run_chain(iters, burn, thin, numGibbs, numMH, seed, 1.0, 0.5, 0.8, splitmerge=mode)
            ch["final"] = ctx.get_state()[0]; ch["cocluster"] = ctx.cocluster(max(ch["num_samples"], 1))
            outs.append(ch); ctx.close()
        a, b = outs
        same = all(np.array_equal(a[f], b[f]) for f in ("clusts", "K", "r", "p", "loglik", "logposterior", "r_acceptances", "splitmerge_acceptances",
                                                         "splitmerge_splits", "r_all", "p_all", "final", "cocluster", "r_final", "p_final", "num_samples"))
        if not same:
            bad += 1
            print("   differing fields:", [f for f in ("clusts", "K", "r", "p", "loglik", "logposterior", "r_acceptances", "splitmerge_acceptances", "splitmerge_splits",
                                                       "r_all", "p_all", "final", "cocluster", "r_final", "p_final", "num_samples") if not np.array_equal(a[f], b[f])],
                  "env", env, "max |dloglik|", float(np.max(np.abs(a["loglik"] - b["loglik"]))) if len(a["loglik"]) else None)
            print(f"CHAIN MISMATCH seed {seed}: n={n} K={K} iters={iters} burn={burn} thin={thin} numGibbs={numGibbs} numMH={numMH} mode={mode} stored={stored} "
                  f"acc {int(a['splitmerge_acceptances'].sum())}/{int(b['splitmerge_acceptances'].sum())} splits {int(a['splitmerge_splits'].sum())}")
    for k, v in saved.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    return bad


def run_chains_oracle(cases, first):
    """rc_run_chain (free-running r / p, split-merge proposals, both modes) against the oracle's loop on random small problems:
    scalar draws, labels, K, split-merge decisions exactly; log-likelihood / log-posterior to 1e-9 relative."""
    bad = 0
    for seed in range(first, first + cases):
        g = np.random.default_rng(seed)
        n = int(g.integers(25, 160)); K = int(g.integers(2, 9)); dim = int(g.integers(max(2, K), K + 4))
        data = rc.generatemixture(n, K, seed=seed, sigma=float(g.uniform(0.3, 0.9)), dim=dim)
        D, truth = data["distancematrix"], data["clusts"]
        P = dict(rc.likelihood_hyperparams(D, truth), maxK=int(g.choice([0, 0, 2 * K])))
        orc = O.Oracle(D, P)
        init = g.integers(1, int(g.integers(1, 2 * K + 1)) + 1, n).astype(np.int64)
        if P["maxK"]: init = (init - 1) % P["maxK"] + 1
        iters, burn, thin = int(g.integers(5, 40)), int(g.integers(0, 5)), int(g.integers(1, 4))
        numGibbs, numMH = int(g.integers(0, 6)), int(g.integers(0, 4))
        intended = bool(g.random() < 0.5); sd = float(g.uniform(0.3, 1.5))
        ctx = rc.Context(D, logD=orc.logD, kcap=n); ctx.set_params(**P); ctx.set_state(init); ctx.cocluster_reset()
        if numMH: ctx.attach_host_matrices(D, orc.logD)
        ch = ctx.run_chain(iters, burn, thin, numGibbs, numMH, seed, 1.0, 0.5, sd, splitmerge="intended" if intended else "as_written")
        ref = O.run_chain(orc, init, 1.0, 0.5, iters, burn, thin, numGibbs, numMH, seed, proposalsd_r=sd, stable=True, intended=intended)
        ok = all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in ((ch["r_all"], ref["r_all"]), (ch["p_all"], ref["p_all"]), (ch["K"], ref["K"]),
                                                                             (ch["r_acceptances"], ref["r_acc"])))
        ok = ok and (len(ref["clusts"]) == 0 or np.array_equal(ch["clusts"], ref["clusts"]))
        if numMH:
            ok = ok and np.array_equal(ch["splitmerge_acceptances"], ref["sm_acc"]) and np.array_equal(ch["splitmerge_splits"], ref["sm_split"])
        if len(ref["loglik"]):
            ok = ok and np.allclose(ch["loglik"], ref["loglik"], rtol=1e-9, atol=0) and np.allclose(ch["logposterior"], ref["logposterior"], rtol=1e-8, atol=0)
        lab = ctx.get_state()[0]
        ok = ok and np.array_equal(lab, orc.clusts)
        if not ok:
            bad += 1
            print(f"ORACLE CHAIN MISMATCH seed {seed}: n={n} K={K} iters={iters} burn={burn} thin={thin} numGibbs={numGibbs} numMH={numMH} intended={intended}")
        ctx.close()
    return bad


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if len(sys.argv) > 3 and sys.argv[3] == "chains":
        bad = run_chains(cases, first)
    elif len(sys.argv) > 3 and sys.argv[3] == "oracle_chains":
        bad = run_chains_oracle(cases, first)
    elif len(sys.argv) > 3 and sys.argv[3] == "wide":      # more than 4096 clusters
        bad = run_wide(cases, first)
    elif len(sys.argv) > 3 and sys.argv[3] == "large":     # beyond the kernel-choice threshold (n > 2560): symmetric kernels, ragged column blocks, re-layouts
        bad = run(cases, first, 2600, 7000, 60)
    else:
        bad = run(cases, first)
    print(f"fuzz: {cases} cases from seed {first}, {bad} bad")
    sys.exit(1 if bad else 0)


