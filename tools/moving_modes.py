"""moving regime (overlapping clusters): sweep time per row-reduction kernel choice and in the exact incremental mode"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, redclust_amd as rc
n, K = 8192, 50
sig = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
d = rc.generatemixture(n, K, seed=2, sigma=sig); D, t = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, t)
for name, bk, mode in (("auto", "auto", 0), ("k_bulk", "perm", 0), ("sym", "sym", 0), ("incremental", "auto", 1)):
    c = rc.Context(D, kcap=512); c.set_params(**P); c.set_state(t)
    c.set_bulk_kernel(bk)
    if mode: c.set_mode("incremental")
    for s in range(60): c.gibbs_sweep(1.0, 0.5, 7, s, blocking=False)
    c.synchronize()
    ch = rd = 0
    t0 = time.perf_counter()
    for s in range(60, 100):
        c.gibbs_sweep(1.0, 0.5, 7, s); st = c.sweep_stats(); ch += st["n_changes"]; rd += st["n_rounds"]
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    for s in range(100, 300): c.gibbs_sweep(1.0, 0.5, 7, s, blocking=False)
    c.synchronize()
    dta = time.perf_counter() - t0
    print(f"{name:12s} sigma {sig}: changes/sweep {ch/40:.1f} rounds/sweep {rd/40:.1f} K {st['K']} blocking {dt/40*1e3:.3f} ms/sweep async {dta/200*1e3:.3f} ms/sweep"
          f" kernel {c.bulk_kernel_name()} layout {c.layout_info()}")
    c.close()
