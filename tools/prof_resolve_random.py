"""Per-phase time of k_resolve for the first sweep from uniformly random labels (profiling build -DRC_PROF_SYML)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
n, K = 8192, 50
d = rc.generatemixture(n, K, seed=1); D, truth = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, truth)
ctx = rc.Context(D, kcap=int(os.environ.get("KCAP", 256))); ctx.set_params(**P)
lab = np.random.default_rng(5).integers(1, K + 1, n)
ctx.set_state(lab); ctx.set_mode("incremental"); ctx.synchronize()
L = rc.lib()
out = np.zeros((8192, 16), np.int64)
L.rc_debug_prof.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
names = {6: "eval tentative", 7: "barrier 1", 8: "assemble batch", 14: "batch_sim", 9: "restore + lists", 10: "eval validate", 11: "barrier 2", 12: "commit"}
if os.environ.get("RC_PROF_COMMIT"): names = {13: "(commit: entries)", 2: "(commit: tables)", 15: "(commit: row sums)", **names}
if os.environ.get("RC_PROF_EVAL"): names = {13: "(valid.: computed)", 15: "(valid.: births+new)", **names}
ctx.gibbs_sweep(1.0, 0.5, 3, 0, blocking=True)
st = ctx.sweep_stats()
L.rc_debug_prof(ctx.h, 0, out.ctypes.data_as(C.c_void_p))
o = out[8192 - 256:, :].astype(np.float64) / 100.0
o = o[o[:, 0] > 0]
print(f"random init: changes {st['n_changes']} rounds {st['n_rounds']}  loop total {np.median(o[:, 4] - o[:, 0]):.1f} us")
print("   entries visited serially:", np.median(out[8192-256:, 15][out[8192-256:, 0] > 0]))
for k, nm in names.items(): print(f"   {nm:18s} {np.median(o[:, k]):9.1f} us   {np.median(o[:, k]) / st['n_rounds']:7.1f} per round   (blocks: min {o[:, k].min():9.1f} max {o[:, k].max():9.1f})")
