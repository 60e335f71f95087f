import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, redclust_amd as rc
n, K = 32768, 200
d = rc.generatemixture(n, K, seed=1, points_only=True)
ctx = rc.Context.from_points(d["points"], kcap=400, storage_bits=32)
t = d["clusts"]
ctx.set_params(delta1=1.0, delta2=1.0, alpha=1.0, beta=1.0, zeta=1.0, gamma=1.0)
P = rc.likelihood_hyperparams_device(ctx, t)
ctx.set_params(**P); ctx.set_state(t)
for s in range(10): ctx.gibbs_sweep(1.0, 0.5, 1, s, blocking=False)
ctx.synchronize()
t0 = time.perf_counter()
for s in range(10, 110): ctx.gibbs_sweep(1.0, 0.5, 1, s, blocking=False)
ctx.synchronize()
dt = time.perf_counter() - t0
print(f"RC_RES_MAXB={os.environ.get('RC_RES_MAXB')}: {100/dt:.0f} sweeps/s ({dt/100*1e3:.3f} ms) {ctx.bulk_kernel_name()} {ctx.sweep_stats()}")
