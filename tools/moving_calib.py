"""label changes per sweep at the moving equilibrium for several cluster overlaps (bench.py's moving-regime sigma)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, redclust_amd as rc
n, K = 8192, 50
for sig in (0.12, 0.14, 0.15, 0.16, 0.18, 0.2):
    d = rc.generatemixture(n, K, seed=2, sigma=sig); D, t = d["distancematrix"], d["clusts"]
    P = rc.likelihood_hyperparams(D, t)
    c = rc.Context(D, kcap=256); c.set_params(**P); c.set_state(t)
    try:
        for s in range(60): c.gibbs_sweep(1.0, 0.5, 7, s, blocking=False)
        c.synchronize()
    except Exception as e:
        print(f"sigma {sig}: {e}"); c.close(); continue
    ch = rd = 0
    t0 = time.perf_counter()
    for s in range(60, 100):
        c.gibbs_sweep(1.0, 0.5, 7, s); st = c.sweep_stats(); ch += st["n_changes"]; rd += st["n_rounds"]
    dt = time.perf_counter() - t0
    print(f"sigma {sig}: changes/sweep {ch/40:.1f} rounds/sweep {rd/40:.1f} K {st['K']}  blocking {dt/40*1e3:.3f} ms/sweep")
    c.close()
