"""End-to-end runsampler throughput (host loop + device sweep + recording) on synthetic data.
usage: python tools/run_e2e.py N K numiters [thin] [init=truth|random]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
N, K, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
thin = int(sys.argv[4]) if len(sys.argv) > 4 else 1
mode = sys.argv[5] if len(sys.argv) > 5 else "truth"
d = rc.generatemixture(N, K, seed=1)
D, truth = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, truth)
params = rc.PriorHyperparamsList(**{k: P[k] for k in ("delta1", "delta2", "alpha", "beta", "zeta", "gamma")})
init = truth if mode == "truth" else np.random.default_rng(0).integers(1, K + 1, size=N).astype(np.int64)
ctx = rc.Context(D, kcap=max(128, 2 * K))
opts = rc.MCMCOptionsList(numiters=iters, burnin=0, thin=thin, numMH=0)
data = rc.MCMCData(D)
t0 = time.perf_counter()
res = rc.runsampler(data, opts, params, rc.MCMCState(init, 1.0, 0.5), verbose=False, seed=1, ctx=ctx)
dt = time.perf_counter() - t0
print(f"N={N} K={K} iters={iters} thin={thin} init={mode}: {iters/dt:.1f} it/s ({dt/iters*1e3:.3f} ms/it), K trace {res.K[:3]}..{res.K[-3:]}, "
      f"loglik {res.loglik[-1]:.3f}, r_acc {res.r_acceptance_rate:.2f}")
# moving regime: sweeps from a random init, per-sweep time and change counts
ctx.set_state(np.random.default_rng(0).integers(1, K + 1, size=N).astype(np.int64))
for t in range(6):
    t0 = time.perf_counter(); ctx.gibbs_sweep(1.0, 0.5, 5, t); dt = time.perf_counter() - t0
    st = ctx.sweep_stats()
    print(f"  random-init sweep {t}: {dt*1e3:.2f} ms, changes {st['n_changes']}, K {st['K']}")
