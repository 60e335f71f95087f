"""Does a short timed window (bench.py --steps 20 --warmup 5) depend on what the GPU did just before?  20 timed sweeps after 5
warm-up sweeps, (a) after a host-side pause, (b) right after 1500 untimed sweeps, (c) after a 2 GiB streaming-read measurement."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
n, K = 8192, 50
d = rc.generatemixture(n, K, seed=1); D, truth = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, truth)
ctx = rc.Context(D); ctx.set_params(**P); ctx.set_state(truth)
d2 = rc.generatemixture(n, K, seed=2, sigma=0.2)
other = rc.Context(d2["distancematrix"], kcap=512); other.set_params(**rc.likelihood_hyperparams(d2["distancematrix"], d2["clusts"])); other.set_state(d2["clusts"])
t = [0]
def window(steps=20, warm=5):
    for _ in range(warm): ctx.gibbs_sweep(1.0, 0.5, 1, t[0], blocking=False); t[0] += 1
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): ctx.gibbs_sweep(1.0, 0.5, 1, t[0], blocking=False); t[0] += 1
    ctx.synchronize()
    return steps / (time.perf_counter() - t0)
for rep in range(3):
    time.sleep(0.5); a = window()
    for _ in range(1500): ctx.gibbs_sweep(1.0, 0.5, 1, t[0], blocking=False); t[0] += 1
    ctx.synchronize(); b = window()
    time.sleep(0.5); rc.measure_read_ceiling(0, 2048, 5); c = window()
    time.sleep(0.5); w200 = window(200, 20)
    time.sleep(0.5)
    for q in range(1500): other.gibbs_sweep(1.0, 0.5, 7, q, blocking=False)
    other.synchronize(); e = window()
    print(f"after a pause {a:.0f} sweeps/s; after 1500 sweeps {b:.0f}; after the read-ceiling measurement {c:.0f}; 200 steps after a pause {w200:.0f}; after 1500 sweeps of ANOTHER context {e:.0f}")
