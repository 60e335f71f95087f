import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import redclust_amd as rc
import oracle_lib as O
bad = 0
for seed, numMH, mode in ((1, 0, "full"), (2, 1, "incremental"), (3, 0, "incremental"), (4, 1, "full")):
    n, K = 4400, 20
    data = rc.generatemixture(n, K, seed=seed, sigma=0.1)
    sh = np.random.default_rng(seed).permutation(n)
    D = np.ascontiguousarray(data["distancematrix"][np.ix_(sh, sh)]); truth = data["clusts"][sh]
    P = dict(rc.likelihood_hyperparams(D, truth), repulsion=False)
    L = np.log(np.where(np.eye(n, dtype=bool), 1.0, D))
    init = np.empty(n, np.int64); init[:300] = 1; init[300:] = np.arange(2, n - 300 + 2)
    ctx = rc.Context(D, logD=L, kcap=4150)
    ctx.set_params(**P); ctx.set_state(init); ctx.set_mode(mode); ctx.cocluster_reset()
    eD, eL = ctx.debug_rowsums(1)[2:4]
    orc = O.Oracle(D, P, logD=L, eL=eL, eD=eD)
    ctx.attach_host_matrices(D, L)
    iters = 4
    rtr = np.full(iters, 1.0); ptr = np.full(iters, 1e-6)
    ch = ctx.run_chain(iters, 0, 1, 2, numMH, 31 + seed, 1.0, 1e-6, 1.0, rp_trace=(rtr, ptr))
    ref = O.run_chain(orc, init, 1.0, 1e-6, iters, 0, 1, 2, numMH, 31 + seed, stable=True, rp_trace=(rtr, ptr))
    ok = (np.array_equal(ch["clusts"], ref["clusts"]) and np.array_equal(ch["K"], ref["K"]) and np.allclose(ch["logposterior"], ref["logposterior"], rtol=1e-9, atol=0)
          and np.array_equal(ch["splitmerge_acceptances"], ref["sm_acc"]))
    print(f"seed {seed} numMH {numMH} mode {mode}: K0 {len(np.unique(init))} K trace {ref['K'].tolist()} device K {ch['K'].tolist()} capacity {ctx.capacity_info()} -> {'ok' if ok else 'MISMATCH'}")
    bad += not ok
    ctx.close()
print("bad", bad)
