"""Compare builds of the library on the headline sweep (N = 8192, K = 50, D only): sweep rate of the pipeline, mean duration of the
row-reduction kernel (HIP events in the dispatch) in the pipeline and alone (blocking sweeps), and a checksum of the exact row-sum
table + the labels after a perturbed start (must be identical across builds: the sums are integers).
usage: python tools/syml_variants.py build_r3/a.so build_r3/b.so ...      (each build runs in its own process: RC_LIB_PATH)
       python tools/syml_variants.py --one       (worker: the build named by RC_LIB_PATH, or the in-tree one)"""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import numpy as np
    import redclust_amd as rc
    n, K = int(os.environ.get("RC_BENCH_N", 8192)), int(os.environ.get("RC_BENCH_K", 50))
    d = rc.generatemixture(n, K, seed=1)
    D, truth = d["distancematrix"], d["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    ctx = rc.Context(D)
    ctx.set_params(**P)
    # exactness: perturbed start, 3 sweeps, labels + row sums of three clusters
    init = truth.copy()
    idx = np.random.default_rng(11).choice(n, n // 50, replace=False)
    init[idx] = np.random.default_rng(12).integers(1, K + 1, size=len(idx))
    ctx.set_state(init)
    h = hashlib.sha256()
    for t in range(3):
        ctx.gibbs_sweep(1.0, 0.5, 8192, t)
    lab = ctx.get_state()[0]
    h.update(lab.tobytes())
    for k in np.unique(lab)[[0, 7, -1]]:
        sd, sl, eD, eL = ctx.debug_rowsums(int(k))
        h.update(sd.tobytes()); h.update(sl.tobytes())
    ctx.set_state(truth)
    sw = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        for _ in range(256):
            ctx.gibbs_sweep(1.0, 0.5, 1, sw, blocking=False); sw += 1
        ctx.synchronize()
    rates, kms = [], []
    for rep in range(5):
        ctx.kernel_timing(enable=1)
        t0 = time.perf_counter()
        for _ in range(200):
            ctx.gibbs_sweep(1.0, 0.5, 1, sw, blocking=False); sw += 1
        ctx.synchronize()
        rates.append(200 / (time.perf_counter() - t0))
        ms, cnt = ctx.kernel_timing(enable=0)
        kms.append(ms / max(cnt, 1))
    # the kernel alone: blocking sweeps (no resolver beside it, no overlap with the next launch)
    ctx.kernel_timing(enable=1)
    for _ in range(100):
        ctx.gibbs_sweep(1.0, 0.5, 1, sw, blocking=True); sw += 1
    ms, cnt = ctx.kernel_timing(enable=0)
    alone = ms / max(cnt, 1)
    rates.sort(); kms.sort()
    print(json.dumps(dict(lib=os.environ.get("RC_LIB_PATH", "in-tree"), sweeps_per_s=rates[2], sweeps_min_max=[rates[0], rates[-1]],
                          kernel_us_pipeline=kms[2] * 1e3, kernel_us_alone=alone * 1e3, kernel=ctx.bulk_kernel_name(), checksum=h.hexdigest()[:16],
                          env={k: v for k, v in os.environ.items() if k.startswith("RC_") and k != "RC_LIB_PATH"})))
    ctx.close()


if __name__ == "__main__":
    if "--one" in sys.argv:
        one()
    else:
        for so in sys.argv[1:] or [""]:
            env = dict(os.environ)
            if so:
                env["RC_LIB_PATH"] = os.path.abspath(so)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
            print(lines[-1] if lines else f'{{"lib": "{so}", "error": {json.dumps(r.stderr.decode()[-400:])}}}')
            sys.stdout.flush()
