"""Per-phase time of k_resolve summed over the rounds of a sweep, moving regime (profiling build -DRC_PROF_SYML):
RC_LIB_PATH=build_exp/lib_prof.so python3 tools/prof_resolve_moving.py [sigma] [kcap] [incremental]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
n, K = 8192, 50
sig = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
kcap = int(sys.argv[2]) if len(sys.argv) > 2 else 0
inc = len(sys.argv) > 3 and sys.argv[3] == "incremental"
d = rc.generatemixture(n, K, seed=2, sigma=sig); D, truth = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, truth)
ctx = rc.Context(D, kcap=kcap); ctx.set_params(**P); ctx.set_state(truth)
if inc: ctx.set_mode("incremental")
for t in range(60): ctx.gibbs_sweep(1.0, 0.5, 7, t, blocking=False)
ctx.synchronize()
L = rc.lib()
out = np.zeros((8192, 16), np.int64)
L.rc_debug_prof.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
names = {6: "eval tentative", 7: "barrier 1", 8: "assemble batch", 14: "batch_sim", 9: "restore + lists", 10: "eval validate", 11: "barrier 2", 12: "commit"}
base0 = None
if os.environ.get("RC_PROF_SIM"):
    L.rc_debug_prof(ctx.h, 0, out.ctypes.data_as(C.c_void_p)); base0 = out[:2].reshape(-1)[:25].astype(np.float64).copy()
if os.environ.get("RC_PROF_COMMIT"): names = {13: "(commit: entries)", 2: "(commit: tables)", 15: "(commit: row sums)", **names}   # -DRC_PROF_COMMIT build: columns 13 / 2 / 15 = entry loop / table rebuild / row-sum corrections inside the commit
if os.environ.get("RC_PROF_EVAL"): names = {2: "(valid.: cached)", 13: "(valid.: computed)", 15: "(valid.: births+new)", **names}   # -DRC_PROF_EVAL build: the longest thread per part of the validation passes
acc = {k: [] for k in names}; mx = {k: [] for k in names}; mn = {k: [] for k in names}; rounds = []; tot = []
for t in range(60, 80):
    ctx.gibbs_sweep(1.0, 0.5, 7, t, blocking=True)
    st = ctx.sweep_stats()
    L.rc_debug_prof(ctx.h, t & 1, out.ctypes.data_as(C.c_void_p))
    o = out[8192 - 256:, :].astype(np.float64) / 100.0
    o = o[o[:, 0] > 0]
    for k in names: acc[k].append(np.median(o[:, k])); mx[k].append(o[:, k].max()); mn[k].append(o[:, k].min())
    rounds.append(st["n_rounds"]); tot.append(np.median(o[:, 4] - o[:, 0]))
    pro = np.median(o[:, 1] - o[:, 0]); epi0 = [round(float(o[b, 5] - o[b, 4]), 1) for b in range(4)]; epi = np.median(o[4:, 5] - o[4:, 4]); span = o[:, 5].max() - o[:, 0].min()
print(f"sigma {sig} kcap {kcap} K {st['K']} rounds/sweep {np.mean(rounds):.1f}  loop total {np.mean(tot):.1f} us (median block)")
print(f"   last sweep: prologue (tables) {pro:.1f} us, epilogue of blocks 0-3 {epi0} us (other blocks {epi:.1f}), first start to last end {span:.1f} us")
for k, nm in names.items(): print(f"   {nm:16s} {np.mean(acc[k]):8.1f} us per sweep   {np.mean(acc[k]) / np.mean(rounds):7.1f} per round   (blocks: min {np.mean(mn[k]):7.1f} max {np.mean(mx[k]):7.1f} per sweep)")
if os.environ.get("RC_PROF_SIM"):   # -DRC_PROF_SIM build: block 0's batch_sim accumulators (cumulative over all sweeps; 10 ns ticks) in row 0 of parity 0
    L.rc_debug_prof(ctx.h, 0, out.ctypes.data_as(C.c_void_p))
    r = out[:2].reshape(-1)[:25].astype(np.float64) - base0; c = r[0]
    print(f"   batch_sim calls {int(c)}: per call {r[10] / c / 100:.1f} us = state set-up {r[1] / c / 100:.1f} + chunk prefetch {r[2] / c / 100:.1f} + entry loop {r[3] / c / 100:.1f} + chunk write-back {r[4] / c / 100:.1f};"
          f" per call {r[9] / c:.0f} entries in {r[8] / c:.1f} chunks, {r[5] / c:.1f} applied one by one ({r[6] / c:.1f} births, {r[7] / c:.1f} deaths): {r[3] / max(r[5], 1) * 10:.0f} ns each")
    if r[15] == 0 and r[13] > 0:   # batch_sim_fast (round 4): [11]/[12] entries decided from the running sizes, [13]/[14] label events (s_memtime ticks)
        print(f"   batch_sim_fast: per call {r[11] / c:.1f} entries decided from the running sizes ({r[12] / max(r[11], 1):.0f} cycles each), {r[13] / c:.1f} label events ({r[14] / max(r[13], 1):.0f} cycles each)")
        sys.exit(0)
    print(f"   serial entries per call by path: renames {r[11] / c:.1f} ({r[12] / max(r[11], 1):.0f} cycles each), certain deaths {r[13] / c:.1f} ({r[14] / max(r[13], 1):.0f} cycles each), "
          f"general {r[15] / c:.1f} ({r[16] / max(r[15], 1):.0f} cycles each); loop iterations {r[17] / c:.1f} per call, preamble {r[18] / max(r[17], 1):.0f} cycles each (s_memtime ticks)")
    print(f"   general path per call: births {r[19] / c:.1f}, deaths {r[20] / c:.1f}, renames {r[21] / c:.1f}, placeholders {r[22] / c:.1f}, no-ops {r[23] / c:.1f}, plain moves {r[24] / c:.1f}")
