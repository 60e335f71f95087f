"""Phase times of k_resolve from a profiling build (-DRC_PROF_SYML): per block, 100 MHz real-time stamps at entry, after the
table set-up, before / after the first grid barrier, at the end of the round loop and at the block's exit."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
n, K = 8192, 50
d = rc.generatemixture(n, K, seed=1); D, truth = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, truth)
ctx = rc.Context(D, kcap=128); ctx.set_params(**P); ctx.set_state(truth)
blocking = bool(int(os.environ.get("BLOCKING", "0")))
for t in range(40): ctx.gibbs_sweep(1.0, 0.5, 1, t, blocking=blocking)
ctx.synchronize()
L = rc.lib()
out = np.zeros((8192, 16), np.int64)
L.rc_debug_prof.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
for gen in (0, 1):
    L.rc_debug_prof(ctx.h, gen, out.ctypes.data_as(C.c_void_p))
    o = out[8192 - 256:, :6].astype(np.float64) / 100.0       # µs
    o = o[o[:, 0] > 0]
    t0 = o[:, 0].min()
    names = ["entry (spread)", "tables ready", "eval done", "barrier passed", "loop left", "exit"]
    print(f"gen {gen}: blocks {len(o)}")
    for k, nm in enumerate(names):
        v = o[:, k] - t0
        print(f"   {nm:16s} median {np.median(v):7.2f}  min {v.min():7.2f}  max {v.max():7.2f} us   (block 0: {o[0, k] - t0:7.2f})")
