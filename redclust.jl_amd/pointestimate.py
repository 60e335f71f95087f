"""Point estimates and clustering comparison — host mirror of /root/reference/src/pointestimate.jl and
src/summaries.jl.  The pairwise loss matrix of the MPEL search and every pair measure are computed on the GPU
(csrc/pointestimate.inc.hip) through the C ABI; there is no CPU fallback."""
from __future__ import annotations

import sys

import numpy as np

from . import _lib

_LOSSES = {"binder": 0, "omARI": 1, "VI": 2, "ID": 3}  # pointestimate.jl:21, 38-47


def _check_lengths(a, b, msg):
    if len(a) != len(b):
        raise ValueError(msg)  # ArgumentError in the reference


def _labels(x):
    """Labels as the reference types them (ClustLabelVector = Vector{Int}); any positive integers are accepted by
    Clustering.jl's counts-based measures, so values above n are compacted first."""
    x = np.asarray(x)
    if x.ndim != 1 or not np.issubdtype(x.dtype, np.integer):
        raise TypeError("cluster labels must be a vector of integers")
    x = x.astype(np.int64)
    if len(x) and (x.min() < 1 or x.max() > len(x)):
        x = np.unique(x, return_inverse=True)[1].astype(np.int64) + 1
    return x


def getpointestimate(samples, method: str = "MAP", loss="VI", device: int = 0):
    """pointestimate.jl:18-60.  Returns (clust, i): a clustering among `samples.clusts` and its sample index —
    1-based as in the reference's return value (samples.clusts[i-1] in Python)."""
    if method == "MPEL" and isinstance(loss, str) and loss not in _LOSSES:
        raise ValueError("Invalid loss function specifier.")
    if method not in ("MAP", "MLE", "MPEL"):
        raise ValueError("Invalid method specifier.")
    if method == "MAP":
        i = int(np.argmax(samples.logposterior))
        return samples.clusts[i], i + 1
    if method == "MLE":
        i = int(np.argmax(samples.loglik))
        return samples.clusts[i], i + 1
    clusts = samples.clusts
    if callable(loss):
        # a user-supplied loss runs where the user's code runs: on the host, pair by pair (pointestimate.jl:36-37,49-58)
        m = len(clusts)
        L = np.zeros((m, m))
        for a in range(m):
            for b in range(a + 1, m):
                L[a, b] = loss(clusts[a], clusts[b])
        L = L + L.T
        i = int(np.argmin(L.sum(axis=0)))
        return clusts[i], i + 1
    S = np.stack([_labels(c) for c in clusts])
    _, _, i, _ = _lib.loss_matrix(S, _LOSSES[loss], device=device, want_matrix=False)
    return clusts[i], i + 1


def lossmatrix(samples, loss: str = "VI", device: int = 0):
    """The symmetrised matrix of pairwise losses that getpointestimate(method="MPEL") searches (pointestimate.jl:49-56)
    and its column sums."""
    clusts = samples.clusts if hasattr(samples, "clusts") else samples
    if loss not in _LOSSES:
        raise ValueError("Invalid loss function specifier.")
    S = np.stack([_labels(c) for c in clusts])
    M, cs, _, _ = _lib.loss_matrix(S, _LOSSES[loss], device=device)
    return M, cs


def binderloss(a, b, normalised: bool = True, device: int = 0) -> float:
    """pointestimate.jl:68-76"""
    _check_lengths(a, b, "Length of the input vectors must be equal.")
    n = len(a)
    pm = _lib.pair_measures(_labels(a), _labels(b), device)
    return pm["mirkin"] * (1 if normalised else n * (n - 1) // 2)


def infodist(a, b, normalised: bool = True, device: int = 0) -> float:
    """pointestimate.jl:89-99"""
    _check_lengths(a, b, "Length of the input vectors must be equal.")
    pm = _lib.pair_measures(_labels(a), _labels(b), device)
    return pm["nid"] if normalised else pm["id"]


def varinfo(a, b, device: int = 0) -> float:
    """Clustering.jl's varinfo, as getpointestimate(loss=varinfo) uses it (test_pointestimates.jl:15)."""
    _check_lengths(a, b, "Length of the input vectors must be equal.")
    return _lib.pair_measures(_labels(a), _labels(b), device)["vi"]


def evaluateclustering(clusts, truth, device: int = 0) -> dict:
    """summaries.jl:12-23 — keys as the reference's named tuple."""
    _check_lengths(clusts, truth, "Length of inputs must be equal.")
    n = len(clusts)
    pm = _lib.pair_measures(_labels(clusts), _labels(truth), device)
    return dict(nbloss=pm["mirkin"], ari=pm["ari"], vi=pm["vi"], nvi=pm["vi"] / np.log(n), id=pm["id"],
                nid=pm["id"] / np.log(n), nmi=pm["nmi"])


def summarise(*args, io=None, device: int = 0) -> None:
    """summaries.jl:32-45: summarise([io], clusts, truth)."""
    if len(args) == 3:
        io, clusts, truth = args
    else:
        clusts, truth = args
    io = io or sys.stdout
    t = evaluateclustering(clusts, truth, device=device)
    print("Clustering summary", file=io)
    print(f"Number of clusters : {len(np.unique(clusts))}", file=io)
    print(f"Normalised Binder loss : {t['nbloss']}", file=io)
    print(f"Adjusted Rand Index : {t['ari']}", file=io)
    print(f"Normalised Variation of Information (NVI) distance : {t['nvi']}", file=io)
    print(f"Normalised Information Distance (NID) : {t['nid']}", file=io)
    print(f"Normalised Mutual Information : {t['nmi']}", file=io)
