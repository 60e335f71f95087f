"""Where does the first (tentative) pass of k_resolve spend its time?  Moving equilibrium (sigma 0.2, N = 8192), profiling build
(-DRC_DIAG -DRC_PROF_SYML): one sweep from the same saved state per ablation (rc_set_option "debug_flags": 4 no Gumbel noise, 16 no
score-cache stores, 1 no candidate loop) — results of such sweeps are wrong on purpose, only the stamps count.
RC_LIB_PATH=build_r4/lib_prof.so python3 tools/exp_tentative.py [incremental]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
n, K = 8192, 50
inc = len(sys.argv) > 1 and sys.argv[1] == "incremental"
d = rc.generatemixture(n, K, seed=2, sigma=0.2); D, truth = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, truth)
ctx = rc.Context(D, kcap=0); ctx.set_params(**P); ctx.set_state(truth)
if inc: ctx.set_mode("incremental")
for t in range(60): ctx.gibbs_sweep(1.0, 0.5, 7, t, blocking=False)
ctx.synchronize()
saved = ctx.get_state()[0].copy()
L = rc.lib(); L.rc_debug_prof.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
out = np.zeros((8192, 16), np.int64)
for flags, name in ((0, "as is"), (4, "no Gumbel noise"), (16, "no cache stores"), (20, "neither"), (0, "as is")):
    vals = []
    for rep in range(3):
        ctx.set_state(saved)
        t = 100 + 2 * rep
        ctx.set_option("debug_flags", 0)
        ctx.gibbs_sweep(1.0, 0.5, 7, t, blocking=True)           # (a normal sweep first: tables, the row sums of the mode, and a changing sweep so that the next one fills the score cache)
        ctx.set_option("debug_flags", flags)
        ctx.gibbs_sweep(1.0, 0.5, 7, t + 1, blocking=True)
        best = None
        for par in (0, 1):                                        # (the parity of the internal sweep index is not exported: take the generation with the later stamps)
            L.rc_debug_prof(ctx.h, par, out.ctypes.data_as(C.c_void_p))
            o = out[8192 - 256:, :].astype(np.float64)
            o = o[o[:, 0] > 0]
            if len(o) and (best is None or o[:, 0].max() > best[:, 0].max()): best = o.copy()
        vals.append(np.median(best[:, 6]) / 100.0)
    ctx.set_option("debug_flags", 0)
    print(f"flags {flags:2d} ({name:18s}): tentative pass {np.median(vals):6.1f} us (runs: {', '.join('%.1f' % v for v in vals)})")
