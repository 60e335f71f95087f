"""Reproduces fuzz case seed 69000 (large): n = 6818, K = 3, sigma = 0.36, 32-bit storage, stored logD, repulsion off, kcap = 76 growing
through 4096 clusters into a wide context.  usage: python tools/repro_wide.py [mode] [bits] [kcap] [maxb] [one_stream]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
mode = sys.argv[1] if len(sys.argv) > 1 else "incremental"
bits_o = int(sys.argv[2]) if len(sys.argv) > 2 else None
kcap_o = int(sys.argv[3]) if len(sys.argv) > 3 else None
os.environ["RC_RES_MAXB"] = sys.argv[4] if len(sys.argv) > 4 else "64"
os.environ["RC_RES_ONE_STREAM"] = sys.argv[5] if len(sys.argv) > 5 else "1"
os.environ["RC_SCORE_CACHE"] = "1"
import redclust_amd as rc
import oracle_lib as O
seed = 69000
g = np.random.default_rng(seed)
n = int(g.integers(2600, 7000)); K = int(g.integers(2, 60)); dim = int(g.integers(max(2, K), K + 6))
sigma = float(g.uniform(0.15, 0.9))
data = rc.generatemixture(n, K, seed=seed, sigma=sigma, dim=dim)
sh = g.permutation(n)
D = np.ascontiguousarray(data["distancematrix"][np.ix_(sh, sh)]); truth = data["clusts"][sh]
P = dict(rc.likelihood_hyperparams(D, truth), repulsion=bool(g.random() < 0.8), maxK=int(g.choice([0, 0, K + 3, 2 * K])))
bits = int(g.choice([64, 64, 32])); stored = bool(g.random() < 0.4) or bits == 32
_ = int(g.choice([512, 512, 64, 16])); _ = int(g.integers(0, 2))
kcap = int(g.choice([min(n, 4096), min(n, 4 * K + 64), 0, 8]))
if bits_o: bits = bits_o; stored = stored or bits == 32
if kcap_o is not None: kcap = kcap_o
print(f"n={n} K={K} sigma={sigma:.2f} bits={bits} stored={stored} kcap={kcap} P.maxK={P['maxK']} rep={P['repulsion']} mode={mode} maxb={os.environ['RC_RES_MAXB']}")
orc0 = O.Oracle(D, P)
ctx = rc.Context(D, logD=orc0.logD if stored else None, kcap=kcap, storage_bits=bits)
ctx.set_params(**P)
L = ctx.get_matrix(1); Dd = ctx.get_matrix(0)
init = g.integers(1, int(g.integers(1, min(n, 3 * K) + 1)) + 1, n).astype(np.int64)
if g.random() < 0.5:
    init = truth.copy(); idx = g.choice(n, max(1, int(n * g.uniform(0.0, 0.05))), replace=False); init[idx] = g.integers(1, K + 1, len(idx))
if P["maxK"]: init = (init - 1) % P["maxK"] + 1
ctx.set_state(init)
eD, eL = ctx.debug_rowsums(int(init[0]))[2:4]
orc = O.Oracle(Dd, P, logD=L, eL=eL, eD=eD)
orc.set_state(init)
_ = g.random()
ctx.set_mode(mode)
_ = g.random()
for t in range(5):
    r, p = float(g.uniform(0.3, 3.0)), float(g.uniform(0.05, 0.95))
    ctx.gibbs_sweep(r, p, seed, t, blocking=bool(t & 1))
    orc.sweep_stable(r, p, seed, t)
    lab, sizes, Kc = ctx.get_state()
    diff = np.flatnonzero(lab != orc.clusts)
    print(f"sweep {t}: device K {Kc} changes {ctx.sweep_stats()['n_changes']} rounds {ctx.sweep_stats()['n_rounds']} | oracle K {orc.K} changes {orc.last_changes} | differing {len(diff)} first at {diff[0] if len(diff) else -1} | capacity {ctx.capacity_info()}")
    if len(diff): break
