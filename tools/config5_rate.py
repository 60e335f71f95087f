"""BASELINE config 5 on one GPU (bench.py's other_configs leg alone, for rocprofv3): N = 32768, K = 200, 32-bit storage, built from the
points on the device; sweeps from the generating labels.  usage: python3 tools/config5_rate.py [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import redclust_amd as rc
n, K = 32768, 200
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
d = rc.generatemixture(n, K, seed=1, points_only=True)
ctx = rc.Context.from_points(d["points"], storage_bits=32)
P = rc.likelihood_hyperparams_device(ctx, d["clusts"])
ctx.set_params(**P); ctx.set_state(d["clusts"])
if os.environ.get("MODE"): ctx.set_mode(os.environ["MODE"])   # MODE=incremental: what runsampler uses (no row reduction per sweep)
if os.environ.get("ABL"): ctx.set_option("debug_flags", int(os.environ["ABL"]))   # (diag builds: timing ablations, wrong sums)
sw = 0
for _ in range(10):
    ctx.gibbs_sweep(1.0, 0.5, 1, sw, blocking=False); sw += 1
ctx.synchronize()
if not os.environ.get("RC_BENCH_NO_TIMING"): ctx.kernel_timing(enable=1)
t0 = time.perf_counter()
for _ in range(steps):
    ctx.gibbs_sweep(1.0, 0.5, 1, sw, blocking=False); sw += 1
ctx.synchronize()
dt = time.perf_counter() - t0
ms, cnt = ctx.kernel_timing(enable=0)
fam, nbytes = ctx.bulk_kernel_info()
print(json.dumps(dict(n=n, K=K, steps=steps, sweeps_per_s=steps / dt, ms_per_sweep=dt / steps * 1e3, kernel=ctx.bulk_kernel_name(), bytes_read_per_launch=nbytes,
                      avg_launch_ms=ms / max(cnt, 1), frac_of_8TBps=(nbytes / (ms / max(cnt, 1) * 1e-3) / 8e12) if cnt else None, stats=ctx.sweep_stats(), capacity=ctx.capacity_info())))
