"""First sweeps from uniformly random labels (bench.py's uniform_init leg alone: N = 8192, K = 50, sigma = 0.1): time, label changes
and resolver rounds of each, and a checksum of the labels reached.  usage: [RC_LIB_PATH=...] python tools/uniform_init.py"""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
n, K = 8192, 50
d = rc.generatemixture(n, K, seed=1); D, truth = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, truth)
ctx = rc.Context(D); ctx.set_params(**P)
out = []
for rep in range(3):
    ctx.set_state(np.random.default_rng(5).integers(1, K + 1, n).astype(np.int64))
    ctx.synchronize()
    rows = []
    for t in range(3):
        t0 = time.perf_counter()
        ctx.gibbs_sweep(1.0, 0.5, 3, t, blocking=True)
        ms = 1e3 * (time.perf_counter() - t0)
        st = ctx.sweep_stats()
        rows.append(dict(ms=round(ms, 3), changes=st["n_changes"], rounds=st["n_rounds"], K=st["K"]))
    out.append(rows)
print(json.dumps(dict(lib=os.environ.get("RC_LIB_PATH", "in-tree"), first_sweeps=out[-1], first_sweep_ms=[o[0]["ms"] for o in out],
                      checksum=hashlib.sha256(ctx.get_state()[0].tobytes()).hexdigest()[:16])))
