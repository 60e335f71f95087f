"""Sweep rate in the moving regime (bench.py's moving_regime leg alone: N = 8192, K = 50, sigma = 0.2, 60 burn-in sweeps) and the
state reached, as a checksum (identical across builds and settings: the chain is exact).  usage: [RC_LIB_PATH=...] python tools/moving_rate.py [sigma]"""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
n, K = 8192, 50
sig = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
d = rc.generatemixture(n, K, seed=2, sigma=sig); D, truth = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, truth)
ctx = rc.Context(D); ctx.set_params(**P); ctx.set_state(truth)
if os.environ.get("MODE"): ctx.set_mode(os.environ["MODE"])        # MODE=incremental: what runsampler uses
sw = 0
for _ in range(60):
    ctx.gibbs_sweep(1.0, 0.5, 7, sw, blocking=False); sw += 1
ctx.synchronize()
ch = rounds = 0
t0 = time.perf_counter()
for _ in range(100):
    ctx.gibbs_sweep(1.0, 0.5, 7, sw, blocking=True); sw += 1
    st = ctx.sweep_stats(); ch += st["n_changes"]; rounds += st["n_rounds"]
t_block = time.perf_counter() - t0
rates, kus = [], []
for rep in range(3):
    ctx.kernel_timing(enable=1)
    t0 = time.perf_counter()
    for _ in range(200):
        ctx.gibbs_sweep(1.0, 0.5, 7, sw, blocking=False); sw += 1
    ctx.synchronize()
    rates.append(200 / (time.perf_counter() - t0))
    ms, cnt = ctx.kernel_timing(enable=0)
    kus.append(1e3 * ms / max(cnt, 1))
lab = ctx.get_state()[0]
print(json.dumps(dict(reduction_us=sorted(kus)[1], kernel=ctx.bulk_kernel_name(), mode=os.environ.get("MODE", "full"), lib=os.environ.get("RC_LIB_PATH", "in-tree"), sigma=sig, sweeps_per_s=sorted(rates)[1], rates=rates, blocking_sweeps_per_s=100 / t_block,
                      changes_per_sweep=ch / 100, rounds_per_sweep=rounds / 100, K=ctx.sweep_stats()["K"],
                      checksum=hashlib.sha256(lab.tobytes()).hexdigest()[:16], capacity=ctx.capacity_info())))
