"""Where do split-merge proposals get accepted?  (bench.py needs a leg in which the speculative pipeline rolls back.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
for n, K, sig in ((2000, 20, 0.3), (2000, 20, 0.4), (8192, 50, 0.3), (2000, 8, 0.5)):
    d = rc.generatemixture(n, K, seed=3, sigma=sig); D, truth = d["distancematrix"], d["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    ctx = rc.Context(D); ctx.set_params(**P); ctx.set_state(truth); ctx.cocluster_reset(); ctx.attach_host_matrices(D)
    for mode in ("as_written", "intended"):
        t0 = time.perf_counter()
        ch = ctx.run_chain(300, 0, 10, 5, 1, 5, 1.0, 0.5, 1.0, splitmerge=mode)
        dt = time.perf_counter() - t0
        print(n, K, sig, mode, "acc", int(ch["splitmerge_acceptances"].sum()), "splits", int(ch["splitmerge_splits"].sum()), "K", int(ch["K"][-1]), "it/s", round(300 / dt), ctx.chain_stats())
    ctx.close()
