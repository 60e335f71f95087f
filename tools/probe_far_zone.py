"""Finds a small moving-regime configuration whose equilibrium lies between K = n/64 and K = n/36 with more than n/32 label runs (the zone
in which the automatic re-layout runs with a backed-off interval).  usage: python tools/probe_far_zone.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
for n, K0, sig in [(4096, 20, 0.2), (4096, 30, 0.2), (4096, 40, 0.18), (3072, 20, 0.2), (3072, 30, 0.18), (4096, 25, 0.22)]:
    d = rc.generatemixture(n, K0, seed=4, sigma=sig); D, truth = d["distancematrix"], d["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    ctx = rc.Context(D); ctx.set_params(**P); ctx.set_state(truth)
    tr = []
    for t in range(400):
        ctx.gibbs_sweep(1.0, 0.5, 9, t, blocking=False)
        if t % 40 == 39:
            st = ctx.sweep_stats(); li = ctx.layout_info()
            tr.append((t + 1, st["K"], li[1], li[0], st["n_changes"], ctx.bulk_kernel_name()[:12]))
    print(f"n={n} K0={K0} sigma={sig}: n/64={n // 64} n/36={n // 36} n/32={n // 32}  (sweep, K, runs, layouts, changes, kernel):", tr)
    ctx.close()
