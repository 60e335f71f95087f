"""Per-round record of the first sweep from uniformly random labels (trace build -DRC_TRACE_RESOLVE), N = 8192, K = 50.
RC_LIB_PATH=build_r3/lib_trace.so python3 tools/trace_uniform.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
n, K = 8192, 50
d = rc.generatemixture(n, K, seed=1); D, truth = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, truth)
ctx = rc.Context(D); ctx.set_params(**P)
ctx.set_state(np.random.default_rng(5).integers(1, K + 1, n).astype(np.int64))
L = rc.lib()
out = np.zeros((8192, 16), np.int64)
L.rc_debug_prof.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
ctx.gibbs_sweep(1.0, 0.5, 3, 0, blocking=True)
st = ctx.sweep_stats()
L.rc_debug_prof(ctx.h, 0, out.ctypes.data_as(C.c_void_p))
tr = out.reshape(-1)[:960].reshape(120, 8)
print(f"changes {st['n_changes']} rounds {st['n_rounds']} K {st['K']}")
for q, r in enumerate(tr[:40]):
    if r[0] != q: break
    print("   round %d: announced %d kept %d hi %d first %d after %d effective %d births %d exact %d" % (*r[:7], r[7] & 0xFFFF, r[7] >> 16))
