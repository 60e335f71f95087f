"""Shared helpers for the tests (golden loading, synthetic cases)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
NAMES = ("delta1", "delta2", "alpha", "beta", "zeta", "gamma", "eta", "sigma", "u", "v")


def load_golden():
    return np.load(os.path.join(GOLD, "golden_sweeps.npz")), np.load(os.path.join(GOLD, "paper_datasets.npz"))


def golden_case(g, d, tag):
    ds = int(tag[1])
    P = dict(zip(NAMES, (float(x) for x in g[f"{tag}_params"])))
    P["repulsion"] = bool(g[f"{tag}_repulsion"])
    P["maxK"] = int(g[f"{tag}_maxK"])
    return d[f"D{ds}"], P, g[f"{tag}_init"].astype(np.int64), int(g[f"{tag}_seed"])


def rp_schedule(t):
    return 0.7 + 0.37 * ((t * 7) % 5), 0.15 + 0.1 * ((t * 3) % 7)


def assert_derived_log_close(L, D, eD, eL, block=1024):
    """Per-entry bound of DESIGN.md §3 on the device's derived logD against libm's log of the host's D:
    |L - log D| <= 2^-eL (rounding to the quantum of logD) + 2.6e-13 (degree-4 polynomial of log1p, |r| <= 1/257)
                   + 2^-(eD+1) / D[i,j] (relative rounding of the fixed-point entry the log is taken of),
    off the diagonal; the diagonal is exactly 0 (types.jl:155).  Row blocks keep the temporaries small at n = 8192.
    Returns (largest error, largest error / bound)."""
    import numpy as np
    n = D.shape[0]
    assert np.all(np.diag(L) == 0.0)
    worst, worst_ratio = 0.0, 0.0
    for a in range(0, n, block):
        b = min(n, a + block)
        Db = D[a:b].copy()
        Db[np.arange(b - a), np.arange(a, b)] = 1.0
        err = np.abs(L[a:b] - np.log(Db))
        bound = 2.0 ** -eL + 2.6e-13 + 2.0 ** -(eD + 1) / Db
        err[np.arange(b - a), np.arange(a, b)] = 0.0
        ratio = err / bound
        worst = max(worst, float(err.max())); worst_ratio = max(worst_ratio, float(ratio.max()))
    assert worst_ratio <= 1.0, (worst, worst_ratio, eD, eL)
    return worst, worst_ratio
