"""Plain sweeps in the moving regime (sigma 0.2, K ~ 206, ~40 label changes per sweep), full mode, asynchronous — the workload of
bench.py's moving_regime block, for kernel traces (tools/kt_moving.sh).  usage: python3 tools/moving_sweeps.py [sigma] [kcap] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
n, K = 8192, 50
sig = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
kcap = int(sys.argv[2]) if len(sys.argv) > 2 else 512
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
d = rc.generatemixture(n, K, seed=2, sigma=sig); D, truth = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, truth)
ctx = rc.Context(D, kcap=kcap); ctx.set_params(**P); ctx.set_state(truth)
for t in range(60): ctx.gibbs_sweep(1.0, 0.5, 7, t, blocking=False)
ctx.synchronize()
t0 = time.perf_counter()
ch = 0
for t in range(60, 60 + steps): ctx.gibbs_sweep(1.0, 0.5, 7, t, blocking=False)
ctx.synchronize()
dt = time.perf_counter() - t0
print(f"sigma {sig}: {steps / dt:.0f} sweeps/s ({dt / steps * 1e6:.0f} us per sweep), last sweep {ctx.sweep_stats()}")
