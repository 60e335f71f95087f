"""One-off randomised check: rc_run_chain (r / p updates, split-merge proposals, sweeps, recording) in WIDE contexts (more than 4096
clusters) against the oracle's loop — the shape of tests/test_gpu_wide.py::test_chain_in_a_wide_context with random sizes, seeds,
capacities (incl. wide-but-not-n, so that the chain can overflow a wide context), numMH and modes.  usage: python tools/fuzz_wide_chains.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import redclust_amd as rc
import oracle_lib as O
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 6
first = int(sys.argv[2]) if len(sys.argv) > 2 else 95000
bad = 0
for seed in range(first, first + cases):
    g = np.random.default_rng(seed)
    n = int(g.integers(4150, 4700)); K = int(g.integers(5, 30))
    data = rc.generatemixture(n, K, seed=seed, sigma=float(g.uniform(0.08, 0.3)))
    D, truth = data["distancematrix"], data["clusts"]
    P = dict(rc.likelihood_hyperparams(D, truth), repulsion=bool(g.random() < 0.3))
    L = np.log(np.where(np.eye(n, dtype=bool), 1.0, D))
    init = np.arange(1, n + 1, dtype=np.int64)
    merged = g.choice(n, int(g.integers(0, n - 4120)), replace=False)
    if len(merged): init[merged] = init[merged[0]]
    kcap = int(g.choice([0, 4096, int(g.integers(4097, n)), n]))
    numMH = int(g.integers(0, 2)); iters = int(g.integers(3, 6)); mode = "incremental" if g.random() < 0.5 else "full"
    p0 = float(g.choice([1e-6, 1e-3, 0.2]))
    try:
        ctx = rc.Context(D, logD=L, kcap=kcap)
        ctx.set_params(**P); ctx.set_state(init); ctx.set_mode(mode); ctx.cocluster_reset()
        eD, eL = ctx.debug_rowsums(int(init[0]))[2:4]
        orc = O.Oracle(D, P, logD=L, eL=eL, eD=eD)
        ctx.attach_host_matrices(D, L)
        rtr = np.full(iters, 1.0); ptr = np.full(iters, p0)
        ch = ctx.run_chain(iters, 0, 1, 2, numMH, seed, 1.0, p0, 1.0, rp_trace=(rtr, ptr))
        ref = O.run_chain(orc, init, 1.0, p0, iters, 0, 1, 2, numMH, seed, stable=True, rp_trace=(rtr, ptr))
        ok = (np.array_equal(ch["clusts"], ref["clusts"]) and np.array_equal(ch["K"], ref["K"]) and
              np.array_equal(ch["splitmerge_acceptances"], ref["sm_acc"]) and np.array_equal(ch["splitmerge_splits"], ref["sm_split"]) and
              np.allclose(ch["logposterior"], ref["logposterior"], rtol=1e-9, atol=0))
        print(f"seed {seed}: n={n} K0={len(np.unique(init))} kcap={kcap} numMH={numMH} iters={iters} mode={mode} p={p0} rep={P['repulsion']} K trace {ref['K'].tolist()} capacity {ctx.capacity_info()['kcap']} -> {'ok' if ok else 'MISMATCH'}")
        ctx.close()
    except rc.RedClustHIPError as e:
        ok = False; print(f"seed {seed}: ERROR {e}")
    bad += not ok
print(f"wide chains: {cases} cases from seed {first}, {bad} bad")
