"""Per-round record of one resolver block (trace build -DRC_TRACE_RESOLVE): changers announced, entries kept, effective, births
RC_LIB_PATH=build_exp/lib_trace.so python3 tools/trace_resolve.py [sigma] [kcap]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
n, K = 8192, 50
sig = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
kcap = int(sys.argv[2]) if len(sys.argv) > 2 else 0
d = rc.generatemixture(n, K, seed=2, sigma=sig); D, truth = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, truth)
ctx = rc.Context(D, kcap=kcap); ctx.set_params(**P); ctx.set_state(truth); ctx.set_mode("incremental")
for t in range(60): ctx.gibbs_sweep(1.0, 0.5, 7, t, blocking=False)
ctx.synchronize()
L = rc.lib()
out = np.zeros((8192, 16), np.int64)
L.rc_debug_prof.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
for t in range(60, 68):
    ctx.gibbs_sweep(1.0, 0.5, 7, t, blocking=True)
    st = ctx.sweep_stats()
    L.rc_debug_prof(ctx.h, t & 1, out.ctypes.data_as(C.c_void_p))
    tr = out.reshape(-1)[:960].reshape(120, 8)
    print(f"sweep {t}: changes {st['n_changes']} rounds {st['n_rounds']} K {st['K']}")
    for q, r in enumerate(tr):
        if r[0] != q: break
        print("   round %d: announced %d kept %d hi %d first %d after %d effective %d births %d" % tuple(r))
sizes = np.bincount(ctx.get_state()[0])[1:]
print("singletons:", int((sizes == 1).sum()), "clusters of size<=3:", int(((sizes > 0) & (sizes <= 3)).sum()))
