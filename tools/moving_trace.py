"""Moving regime in full mode, sweep by sweep: label runs, the row-reduction kernel chosen and the sweep time (blocking), then the
pipelined rate.  usage: [RC_SYM_RUNS_DIV=..] python tools/moving_trace.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
n, K = 8192, 50
d = rc.generatemixture(n, K, seed=2, sigma=0.2); D, truth = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, truth)
ctx = rc.Context(D); ctx.set_params(**P); ctx.set_state(truth)
sw = 0
for _ in range(60):
    ctx.gibbs_sweep(1.0, 0.5, 7, sw, blocking=False); sw += 1
ctx.synchronize()
rows = []
for _ in range(60):
    t0 = time.perf_counter(); ctx.gibbs_sweep(1.0, 0.5, 7, sw, blocking=True); sw += 1
    dt = time.perf_counter() - t0
    rows.append((ctx.layout_info()[1], ctx.bulk_kernel_name()[:14], round(dt * 1e6)))
from collections import Counter
print("kernels", Counter(r[1] for r in rows), "runs min/med/max", min(r[0] for r in rows), sorted(r[0] for r in rows)[30], max(r[0] for r in rows), "relayouts", ctx.layout_info()[0])
for k in sorted(set(r[1] for r in rows)):
    ts = sorted(r[2] for r in rows if r[1] == k)
    print("  ", k, "blocking sweep us median", ts[len(ts) // 2])
rates = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(200):
        ctx.gibbs_sweep(1.0, 0.5, 7, sw, blocking=False); sw += 1
    ctx.synchronize(); rates.append(200 / (time.perf_counter() - t0))
print("pipelined sweeps/s", ["%.0f" % r for r in rates], "div", os.environ.get("RC_SYM_RUNS_DIV", "32"))
