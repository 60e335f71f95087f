"""Times rc_loss_matrix (MPEL loss matrix, pointestimate.jl:49-58) on samples of a real chain and the C oracle on a
bounded number of pairs.   python tools/bench_pointestimate.py [n K m]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import redclust_amd as rc  # noqa: E402
from redclust_amd import _lib  # noqa: E402
import oracle_lib as O  # noqa: E402

n, K, m = (int(x) for x in (sys.argv[1:4] or (8192, 50, 1000)))
sigma = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
data = rc.generatemixture(n, K, seed=1, sigma=sigma)
D, truth = data["distancematrix"], data["clusts"]
P = rc.likelihood_hyperparams(D, truth)
ctx = rc.Context(D, kcap=max(128, 4 * K))
ctx.set_params(**P)
ctx.set_state(truth)
S = np.empty((m, n), np.int64)
for t in range(m):
    ctx.gibbs_sweep(1.0, 0.5, 1, t, blocking=False)
    S[t] = ctx.get_state()[0]
ctx.close()
print("distinct samples:", len({s.tobytes() for s in S}), "K range", min(len(np.unique(s)) for s in S), max(len(np.unique(s)) for s in S))
out = {"n": n, "K": K, "m": m, "pairs": m * (m - 1) // 2}
for loss, kind in (("binder", 0), ("VI", 2)):
    _lib.loss_matrix(S[:4], kind)
    t0 = time.perf_counter()
    M, cs, i, ms = _lib.loss_matrix(S, kind)
    wall = time.perf_counter() - t0
    out[loss] = {"kernel_ms": ms, "wall_s": wall, "pairs_per_s_kernel": out["pairs"] / (ms * 1e-3), "argmin": i,
                 "label_bytes_per_s_kernel": out["pairs"] * 2 * n * 2 / (ms * 1e-3)}
# CPU: the oracle on a few pairs
k = min(m, 12)
t0 = time.perf_counter()
O.mpel(S[:k], 2)
dt = time.perf_counter() - t0
out["cpu_oracle_pairs_per_s"] = (k * (k - 1) // 2) / dt
print(json.dumps(out))
