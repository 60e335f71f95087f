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
