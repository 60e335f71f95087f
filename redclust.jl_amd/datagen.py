"""Synthetic benchmark inputs with the distribution of the reference's generatemixture
(/root/reference/src/utils.jl:101-147) — the build's own generator (Julia's RNG stream cannot be reproduced;
the O(5000·N²·K) oracle-coclustering loop of utils.jl:130-143 is skipped) — and the likelihood
hyperparameters fitted from a labelling with fitprior's formulas (/root/reference/src/prior.jl:73-75,96-110)."""
from __future__ import annotations

import numpy as np
from scipy.special import digamma, polygamma


def generatemixture(N: int, K: int, *, alpha: float | None = None, dim: int | None = None, radius: float = 1.0,
                    sigma: float = 0.1, seed: int = 0, dtype=np.float64, points_only: bool = False):
    if N < 1:
        raise ValueError("N must be greater than 1.")
    if K < 1 or K > N:
        raise ValueError("K must satisfy 1 ≤ K ≤ N.")
    alpha = float(K) if alpha is None else float(alpha)
    dim = K if dim is None else int(dim)
    if alpha <= 0:
        raise ValueError("α must be positive.")
    if dim < K:
        raise ValueError("dim must be ≥ K.")
    if radius <= 0 or sigma <= 0:
        raise ValueError("radius and σ must be positive.")
    rng = np.random.default_rng(seed)
    probs = rng.dirichlet(np.full(K, alpha))                      # utils.jl:113
    clusts = np.sort(rng.choice(K, size=N, p=probs)) + 1          # utils.jl:114 (sorted labels)
    pts = rng.normal(0.0, sigma, size=(N, dim))                   # utils.jl:123-128
    pts[np.arange(N), clusts - 1] += radius                       # centre k = radius·e_k, utils.jl:117-120
    if points_only:   # the n×n matrix is left to the device (MCMCData(points), rc_create_from_points)
        return dict(points=pts, distancematrix=None, clusts=clusts.astype(np.int64), probs=probs)
    # pairwise Euclidean distances (utils.jl:144-145), built block-row-wise so that N = 32768 needs one N×N array
    sq = np.einsum("ij,ij->i", pts, pts)
    D = np.empty((N, N), dtype=np.float64)
    B = 2048
    for i0 in range(0, N, B):
        i1 = min(N, i0 + B)
        blk = pts[i0:i1] @ pts.T
        blk *= -2.0
        blk += sq[i0:i1, None]
        blk += sq[None, :]
        np.maximum(blk, 0.0, out=blk)
        np.sqrt(blk, out=D[i0:i1])
    # exact symmetry (types.jl:149-151 requires it): mirror the upper triangle, zero diagonal
    for i0 in range(0, N, B):
        i1 = min(N, i0 + B)
        for j0 in range(i0, N, B):
            j1 = min(N, j0 + B)
            if i0 == j0:
                t = D[i0:i1, j0:j1]
                iu = np.triu_indices(i1 - i0, 1)
                t.T[iu] = t[iu]
            else:
                D[j0:j1, i0:i1] = D[i0:i1, j0:j1].T
    np.fill_diagonal(D, 0.0)
    return dict(points=pts, distancematrix=D if dtype == np.float64 else D.astype(dtype), clusts=clusts.astype(np.int64), probs=probs)


def _gamma_shape_mle(mean_x, mean_logx):
    s = np.log(mean_x) - mean_logx
    k = (3 - s + np.sqrt((s - 3) ** 2 + 24 * s)) / (12 * s)
    for _ in range(100):
        k_new = k - (np.log(k) - digamma(k) - s) / (1 / k - polygamma(1, k))
        if abs(k_new - k) < 1e-14 * k:
            return float(k_new)
        k = k_new
    return float(k)


def likelihood_hyperparams(D: np.ndarray, labels: np.ndarray, block: int = 2048) -> dict:
    """δ1, α, β from within-cluster distances A; δ2, ζ, γ from between-cluster distances B (upper triangle)."""
    n = D.shape[0]
    cntA = cntB = 0
    sumA = sumB = slogA = slogB = 0.0
    for i0 in range(0, n, block):
        i1 = min(n, i0 + block)
        sub = D[i0:i1]
        same = labels[i0:i1, None] == labels[None, :]
        upper = np.arange(i0, i1)[:, None] < np.arange(n)[None, :]
        a = sub[same & upper]
        b = sub[(~same) & upper]
        cntA += a.size; cntB += b.size
        sumA += float(a.sum()); sumB += float(b.sum())
        slogA += float(np.log(a).sum()); slogB += float(np.log(b).sum())
    d1 = _gamma_shape_mle(sumA / cntA, slogA / cntA)
    d2 = _gamma_shape_mle(sumB / cntB, slogB / cntB)
    return dict(delta1=d1, delta2=d2, alpha=cntA * d1, beta=sumA, zeta=cntB * d2, gamma=sumB,
                eta=1.0, sigma=1.0, u=1.0, v=1.0, repulsion=True, maxK=0)


def likelihood_hyperparams_device(ctx, labels) -> dict:
    """likelihood_hyperparams with the within / between sums taken from the device's block sums (rc_within_between):
    no host pass over the n×n matrix.  ctx: a Context holding D; its state is set to `labels`."""
    ctx.set_state(np.asarray(labels, dtype=np.int64))
    w = ctx.within_between()
    cntA, cntB = w["count_within"], w["count_between"]
    d1 = _gamma_shape_mle(w["sum_within"] / cntA, w["sumlog_within"] / cntA)
    d2 = _gamma_shape_mle(w["sum_between"] / cntB, w["sumlog_between"] / cntB)
    return dict(delta1=d1, delta2=d2, alpha=cntA * d1, beta=w["sum_within"], zeta=cntB * d2, gamma=w["sum_between"],
                eta=1.0, sigma=1.0, u=1.0, v=1.0, repulsion=True, maxK=0)
