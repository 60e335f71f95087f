import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, redclust_amd as rc
N, K = 8192, 50
d = rc.generatemixture(N, K, seed=1); D, truth = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, truth)
params = rc.PriorHyperparamsList(**{k: P[k] for k in ("delta1", "delta2", "alpha", "beta", "zeta", "gamma")})
ctx = rc.Context(D, kcap=128)
opts = rc.MCMCOptionsList(numiters=100, burnin=0, thin=1, numMH=0)
data = rc.MCMCData(D)
pr = cProfile.Profile(); pr.enable()
res = rc.runsampler(data, opts, params, rc.MCMCState(truth, 1.0, 0.5), verbose=False, seed=1, ctx=ctx)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
