"""Host-side mirror of the reference's option / parameter / state / data / result structs
(/root/reference/src/types.jl).  Same names, same defaults, same validation messages."""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np


class MCMCOptionsList:
    """src/types.jl:26-58."""

    def __init__(self, numiters: int = 5000, burnin: int | None = None, thin: int = 1, numGibbs: int = 5,
                 numMH: int = 1):
        if burnin is None:
            burnin = int(math.floor(0.2 * numiters))  # types.jl:35
        if numiters < 1:
            raise ValueError("numiters must be ≥ 1.")
        if burnin > numiters:
            raise ValueError("burnin must be < numiters")
        if thin < 1:
            raise ValueError("thin must be positive.")
        if numGibbs < 0:
            raise ValueError("numGibbs must be non-negative.")
        if numMH < 0:
            raise ValueError("numMH must be non-negative.")
        self.numiters, self.burnin, self.thin, self.numGibbs, self.numMH = int(numiters), int(burnin), int(thin), int(numGibbs), int(numMH)
        self.numsamples = int(math.floor((numiters - burnin) / thin))  # types.jl:55

    def __repr__(self):  # types.jl:60-67
        pl = lambda k: "" if k == 1 else "s"
        return (f"MCMC Options\n{self.numiters} iteration{pl(self.numiters)}\n{self.burnin} burnin iteration{pl(self.burnin)}\n"
                f"{self.numsamples} sample{pl(self.numsamples)}\n{self.numGibbs} restricted Gibbs step{pl(self.numGibbs)} per split-merge step\n"
                f"{self.numMH} split-merge step{pl(self.numMH)} per iteration\n")


@dataclass
class PriorHyperparamsList:
    """src/types.jl:93-108 (ASCII field names: δ1→delta1, δ2→delta2, α→alpha, β→beta, ζ→zeta, γ→gamma,
    η→eta, σ→sigma)."""
    delta1: float = 1.0
    delta2: float = 1.0
    alpha: float = 1.0
    beta: float = 1.0
    zeta: float = 1.0
    gamma: float = 1.0
    eta: float = 1.0
    sigma: float = 1.0
    proposalsd_r: float | None = None
    u: float = 1.0
    v: float = 1.0
    K_initial: int = 1
    repulsion: bool = True
    maxK: int = 0

    def __post_init__(self):
        if self.proposalsd_r is None:
            self.proposalsd_r = math.sqrt(self.eta) / self.sigma  # types.jl:102

    def as_dict(self):
        return {k: getattr(self, k) for k in ("delta1", "delta2", "alpha", "beta", "zeta", "gamma", "eta", "sigma",
                                              "u", "v", "repulsion", "maxK")}


class MCMCState:
    """src/types.jl:131-137: clustsizes = counts(clusts, 1:n), K = sum(clustsizes .> 0)."""

    def __init__(self, clusts, r: float, p: float, clustsizes=None, K=None):
        self.clusts = np.ascontiguousarray(clusts, dtype=np.int64).copy()
        n = len(self.clusts)
        if self.clusts.min() < 1 or self.clusts.max() > n:
            raise ValueError("cluster labels must lie in 1..n")
        self.r, self.p = float(r), float(p)
        self.clustsizes = (np.bincount(self.clusts, minlength=n + 1)[1:].astype(np.int64)
                           if clustsizes is None else np.asarray(clustsizes, dtype=np.int64))
        self.K = int(np.sum(self.clustsizes > 0)) if K is None else int(K)


class MCMCData:
    """src/types.jl:145-162.  D is stored as given; logD = log.(D - Diagonal(D) + I) is computed lazily on the
    host only if somebody reads it — the device derives its own copy (rc_create)."""

    def __init__(self, D_or_points):
        x = D_or_points
        self.points = None
        self._D = None
        self._logD = None
        if isinstance(x, (list, tuple)) or (isinstance(x, np.ndarray) and x.ndim == 2 and x.shape[0] != x.shape[1]):
            # points (one observation per row): the device computes the distances (rc_create_from_points); the host
            # matrix is only materialised if somebody reads .D (e.g. the split–merge step)
            self.points = np.ascontiguousarray(x, dtype=np.float64)
            if self.points.ndim != 2:
                raise ValueError("points must be a vector of equal-length vectors")
            return
        D = np.asarray(x, dtype=np.float64)
        if D.ndim != 2 or D.shape[0] != D.shape[1]:
            raise ValueError("D must be a square matrix.")
        if np.any(D != D.T):
            raise ValueError("D must be symmetric.")
        # Zero (or negative) off-diagonal distances.  The reference accepts them silently: log.(D - Diagonal(D) + I)
        # (types.jl:155) is -Inf there, every log-weight that touches such a pair becomes -Inf or NaN (mcmc.jl:223-247,
        # 0 * -Inf when δ1 = 1) and sample_logweights (utils.jl:2-6) then draws from NaNs.  This implementation refuses the
        # input instead (rc_create returns RC_ERR_DOMAIN for the same reason): DESIGN.md "Zero distances".
        bad = int(np.count_nonzero(D <= 0)) - int(np.count_nonzero(np.diagonal(D) <= 0))
        if bad or not np.all(np.isfinite(D)):
            raise ValueError(f"D must be finite with positive off-diagonal entries ({bad} are zero or negative: log D = -Inf "
                             "there, which the reference would propagate as -Inf / NaN log-weights).  Remove duplicate "
                             "observations or add a small jitter to them.")
        self._D = np.ascontiguousarray(D)

    @property
    def n(self):
        return self.points.shape[0] if self._D is None else self._D.shape[0]

    @property
    def D(self):
        if self._D is None:  # pairwise(Euclidean(), makematrix(pnts), dims=2), types.jl:159-162
            pts = self.points
            sq = np.einsum("ij,ij->i", pts, pts)
            D2 = sq[:, None] + sq[None, :] - 2 * (pts @ pts.T)
            np.maximum(D2, 0.0, out=D2)
            D = np.sqrt(D2)
            D = np.minimum(D, D.T)
            np.fill_diagonal(D, 0.0)
            self._D = np.ascontiguousarray(D)
        return self._D

    @property
    def logD(self):
        if self._logD is None:
            M = self.D.copy()
            np.fill_diagonal(M, 1.0)
            with np.errstate(divide="ignore"):
                self._logD = np.log(M)
        return self._logD

    def __repr__(self):
        return f"MCMC data : {self.n}×{self.n} dissimilarity matrix."


@dataclass
class MCMCResult:
    """src/types.jl:193-248 — every field of the reference's result container."""
    options: MCMCOptionsList = None
    params: PriorHyperparamsList = None
    clusts: list = field(default_factory=list)
    posterior_coclustering: np.ndarray = None
    K: np.ndarray = None
    K_ess: float = 0.0
    K_acf: np.ndarray = None
    K_iac: float = 0.0
    K_mean: float = 0.0
    K_variance: float = 0.0
    r: np.ndarray = None
    r_ess: float = 0.0
    r_acf: np.ndarray = None
    r_iac: float = 0.0
    r_mean: float = 0.0
    r_variance: float = 0.0
    p: np.ndarray = None
    p_ess: float = 0.0
    p_acf: np.ndarray = None
    p_iac: float = 0.0
    p_mean: float = 0.0
    p_variance: float = 0.0
    splitmerge_acceptances: np.ndarray = None
    r_acceptances: np.ndarray = None
    r_acceptance_rate: float = 0.0
    splitmerge_splits: np.ndarray = None
    splitmerge_acceptance_rate: float = 0.0
    runtime: float = 0.0
    mean_iter_time: float = 0.0
    loglik: np.ndarray = None
    logposterior: np.ndarray = None

    @classmethod
    def allocate(cls, data: MCMCData, options: MCMCOptionsList, params: PriorHyperparamsList):
        n, ns = data.n, options.numsamples
        return cls(options=options, params=params, clusts=[np.zeros(n, np.int64) for _ in range(ns)],
                   posterior_coclustering=None, K=np.zeros(ns, np.int64), K_acf=np.zeros(ns), r=np.zeros(ns),
                   r_acf=np.zeros(ns), p=np.zeros(ns), p_acf=np.zeros(ns), loglik=np.zeros(ns),
                   logposterior=np.zeros(ns), splitmerge_acceptances=np.zeros(options.numiters * options.numMH, bool),
                   splitmerge_splits=np.zeros(options.numiters * options.numMH, bool),
                   r_acceptances=np.zeros(options.numiters, bool))

    def __repr__(self):
        return f"MCMC result with {self.options.numsamples} samples"
