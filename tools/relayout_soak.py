"""One-off check of the automatic re-layout between n/64 and n/36 clusters: long full-mode chains in the moving regime against a
context that never re-lays out (forced full-read kernel), states compared every 100 sweeps.  usage: python tools/relayout_soak.py [sweeps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
bad = 0
for n, K0, sig, seed in [(4096, 25, 0.22, 4), (4096, 25, 0.24, 5), (4096, 30, 0.23, 6), (3072, 20, 0.24, 7), (6144, 40, 0.21, 8), (4096, 22, 0.26, 9)]:
    d = rc.generatemixture(n, K0, seed=seed, sigma=sig); D, truth = d["distancematrix"], d["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    ctx = rc.Context(D); ctx.set_params(**P); ctx.set_state(truth)
    ref = rc.Context(D); ref.set_params(**P); ref.set_bulk_kernel("perm"); ref.set_state(truth)
    ok = True; tr = []
    for t in range(sweeps):
        ctx.gibbs_sweep(1.0, 0.5, seed, t, blocking=False); ref.gibbs_sweep(1.0, 0.5, seed, t, blocking=False)
        if t % 100 == 99:
            a, b = ctx.get_state(), ref.get_state()
            ok = ok and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
            li = ctx.layout_info(); tr.append((t + 1, a[2], li[1], li[0], ctx.bulk_kernel_name()[:8]))
    ok = ok and ctx.loglik() == ref.loglik()
    print(f"n={n} K0={K0} sigma={sig}: n/64={n // 64} n/36={n // 36} n/32={n // 32} -> {'ok' if ok else 'MISMATCH'}; (sweep, K, runs, layouts, kernel) {tr[::3]}")
    bad += not ok
    ctx.close(); ref.close()
print("relayout soak:", bad, "bad")
