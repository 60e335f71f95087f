import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
N, K, iters = 8192, 50, 300
d = rc.generatemixture(N, K, seed=1)
D, truth = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, truth)
params = rc.PriorHyperparamsList(**{k: P[k] for k in ("delta1", "delta2", "alpha", "beta", "zeta", "gamma")})
ctx = rc.Context(D, kcap=128)
res = rc.runsampler(rc.MCMCData(D), rc.MCMCOptionsList(numiters=iters, burnin=0, thin=10, numMH=1), params, rc.MCMCState(truth, 1.0, 0.5), verbose=False, seed=1, ctx=ctx)
print("it/s", 1 / res.mean_iter_time, "splits", res.splitmerge_splits.mean(), "acc", res.splitmerge_acceptance_rate)
ctx.close()
