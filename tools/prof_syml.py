"""Per-wave cycle stamps of k_bulk_syml from a profiling build (-DRC_PROF_SYML, RC_LIB_PATH=build_r3/lib_prof.so):
total cycles per wave, cycles parked on the tile's loads (s_waitcnt vmcnt(0) at the top of each tile), tiles, set-up."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
n, K = 8192, 50
d = rc.generatemixture(n, K, seed=1); D, truth = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, truth)
ctx = rc.Context(D); ctx.set_params(**P); ctx.set_state(truth)
blocking = bool(int(os.environ.get("BLOCKING", "0")))
for t in range(40): ctx.gibbs_sweep(1.0, 0.5, 1, t, blocking=blocking)
ctx.synchronize()
L = rc.lib()
out = np.zeros((8192, 16), np.int64)
L.rc_debug_prof.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
for gen in (0, 1):
    L.rc_debug_prof(ctx.h, gen, out.ctypes.data_as(C.c_void_p))
    m = (out[:, 0] > 0) & (out[:, 0] < 10_000_000) & (out[:, 2] > 0) & (out[:, 2] < 1000)   # (the resolver's own stamps share the tail of the buffer)
    o = out[m]
    np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', f'prof_syml_gen{gen}.npy'), out)
    tot, wait, tiles, setup, real, lg, d2, r0 = (o[:, k] for k in range(8))
    iss, ldsw, d1 = o[:, 8], o[:, 9], o[:, 10]
    hw = r0 & 0xFFFFF; r0 = r0 >> 20
    cu = (hw >> 16) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 15)   # xcc, se, sh, cu
    import collections
    cnt = collections.Counter(cu.tolist())
    print('   CUs used', len(cnt), ' waves per CU histogram', sorted(collections.Counter(cnt.values()).items()))
    print(f"gen {gen}: waves {m.sum()}  total cycles/wave median {np.median(tot):.0f} p10 {np.percentile(tot,10):.0f} p90 {np.percentile(tot,90):.0f} max {tot.max()}"
          f" | load-wait share {wait.sum()/tot.sum():.2f}  setup share {setup.sum()/tot.sum():.2f}  tiles/wave {np.median(tiles):.0f}"
          f" | cycles/tile {np.median(tot/np.maximum(tiles,1)):.0f}  wait/tile {np.median(wait/np.maximum(tiles,1)):.0f}"
          f" | clock {np.median(tot/np.maximum(real,1))*100:.0f} MHz  wave lifetime {np.median(real)/100:.1f} us  start spread {(r0.max()-r0.min())/100:.1f} us"
          f" | log share {lg.sum()/tot.sum():.2f}  dir2 share {d2.sum()/tot.sum():.2f}  issue {iss.sum()/tot.sum():.2f}  ldswrite {ldsw.sum()/tot.sum():.2f}  dir1 {d1.sum()/tot.sum():.2f}")
print("(s_memtime ticks: 100 MHz constant clock on gfx9? — compare shares, not absolutes)")
