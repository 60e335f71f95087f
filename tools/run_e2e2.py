"""End-to-end runsampler throughput, native engine vs Python loop.  usage: python tools/run_e2e2.py N K iters"""
import sys, time, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
N, K, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
d = rc.generatemixture(N, K, seed=1)
D, truth = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, truth)
params = rc.PriorHyperparamsList(**{k: P[k] for k in ("delta1", "delta2", "alpha", "beta", "zeta", "gamma")})
ctx = rc.Context(D, kcap=max(128, 2 * K))
data = rc.MCMCData(D)
out = {"n": N, "K": K}
for numMH in (0, 1):
    for thin in (1, 10):
        for eng in ("native", "full"):
            it = iters if numMH == 0 else max(iters // 5, 50)
            opts = rc.MCMCOptionsList(numiters=it, burnin=0, thin=thin, numMH=numMH)
            t0 = time.perf_counter()
            res = rc.runsampler(data, opts, params, rc.MCMCState(truth, 1.0, 0.5), verbose=False, seed=1, ctx=ctx, engine="native", mode="full" if eng == "full" else "incremental")
            dt = time.perf_counter() - t0
            out[f"numMH={numMH} thin={thin} {eng}"] = {"it_per_s_loop": 1.0 / res.mean_iter_time, "it_per_s_call": it / dt}
print(json.dumps(out, indent=1))
