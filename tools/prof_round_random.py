"""Absolute per-block stamps of ONE resolver round (build -DRC_PROF_SYML -DRC_PROF_ROUND=<r>), first sweep from uniformly random labels"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
n, K = 8192, 50
d = rc.generatemixture(n, K, seed=1); D, truth = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, truth)
ctx = rc.Context(D, kcap=256); ctx.set_params(**P); ctx.set_state(np.random.default_rng(5).integers(1, K + 1, n)); ctx.set_mode("incremental")
pass
ctx.synchronize()
L = rc.lib(); out = np.zeros((8192, 16), np.int64)
L.rc_debug_prof.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
ctx.gibbs_sweep(1.0, 0.5, 3, 0, blocking=True)
print(ctx.sweep_stats())
L.rc_debug_prof(ctx.h, 0, out.ctypes.data_as(C.c_void_p))
o = out[8192 - 256:, :].astype(np.float64) / 100.0
names = {6: "tentative pass done", 7: "barrier 1 passed", 8: "batch assembled", 14: "batch_sim done", 9: "lists done", 10: "validation done", 11: "barrier 2 passed", 12: "commit done"}
t0 = o[:, 6][o[:, 6] > 0].min()
for k, nm in names.items():
    v = o[:, k] - t0
    srt = np.argsort(v)
    print(f"{nm:22s} min {v.min():8.1f} median {np.median(v):8.1f} p90 {np.percentile(v, 90):8.1f} max {v.max():8.1f}  latest blocks {list(srt[-4:])}")
