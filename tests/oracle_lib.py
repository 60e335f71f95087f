"""ctypes binding of oracle/librc_oracle.so (TEST INFRASTRUCTURE — see oracle/rc_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ORACLE_DIR, "librc_oracle.so")


class OrcParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("delta1", "delta2", "alpha", "beta", "zeta", "gamma", "eta",
                                          "sigma", "u", "v")] + [("repulsion", C.c_int32), ("maxK", C.c_int64)]


class OrcMHInfo(C.Structure):
    _fields_ = [("accept", C.c_int32), ("split", C.c_int32), ("skipped", C.c_int32), ("pad", C.c_int32),
                ("i", C.c_int64), ("j", C.c_int64), ("nS", C.c_int64), ("log_prior_ratio", C.c_double),
                ("log_lik_ratio", C.c_double), ("log_proposal_ratio", C.c_double), ("log_u", C.c_double)]


def build(force: bool = False) -> str:
    src = os.path.join(ORACLE_DIR, "rc_oracle.c")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return SO


_lib = None
_dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_up = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    PP = C.POINTER(OrcParams)
    i64, f64, u64, i32 = C.c_int64, C.c_double, C.c_uint64, C.c_int
    L.orc_uniform.restype = f64
    L.orc_uniform.argtypes = [u64, u64, u64, u64]
    L.orc_philox.argtypes = [_up, _up, _up]
    L.orc_make_logD.argtypes = [i64, _dp, _dp]
    L.orc_state_from_labels.argtypes = [i64, _ip, _ip, C.POINTER(i64)]
    L.orc_quant_exponent.argtypes = [i64, _dp, i64]
    L.orc_quantize.argtypes = [_dp, i64, i32, _ip]
    L.orc_quant_exponent32.argtypes = [_dp, i64]
    L.orc_sweep_literal.argtypes = [i64, _dp, _dp, _ip, _ip, C.POINTER(i64), PP, f64, f64, u64, u64, i32]
    L.orc_sweep_literal_range.argtypes = [i64, _dp, _dp, _ip, _ip, C.POINTER(i64), PP, f64, f64, u64, u64, i32, i64, i64]
    L.orc_point_scores_literal.restype = i64
    L.orc_point_scores_literal.argtypes = [i64, _dp, _dp, _ip, _ip, PP, f64, f64, i64, _ip, _dp]
    L.orc_size_table.argtypes = [i64, PP, _dp]
    L.orc_sweep_stable.argtypes = [i64, _ip, _ip, i32, i32, _dp, _ip, _ip, C.POINTER(i64), PP, f64, f64,
                                   u64, u64, C.POINTER(i64)]
    L.orc_sweep_table.argtypes = [i64, i64, _ip, _ip, _ip, _ip, i32, i32, _dp, _ip, _ip, C.POINTER(i64), PP, f64, f64, u64, u64,
                                  i64, _ip, _ip, _ip, C.POINTER(i64)]
    L.orc_point_scores_stable.restype = i64
    L.orc_point_scores_stable.argtypes = [i64, _ip, _ip, i32, i32, _dp, _ip, _ip, PP, f64, f64, i64, _ip, _dp]
    L.orc_loglik_literal.restype = f64
    L.orc_loglik_literal.argtypes = [i64, _dp, _dp, _ip, _ip, PP]
    L.orc_loglik_stable.restype = f64
    L.orc_loglik_stable.argtypes = [i64, _ip, _ip, i32, i32, _ip, _ip, PP]
    L.orc_logprior.restype = f64
    L.orc_logprior.argtypes = [i64, _ip, f64, f64, PP]
    L.orc_sortlabels.argtypes = [i64, _ip, _ip]
    L.orc_cocluster_add.argtypes = [i64, _ip, _up]
    L.orc_matsum_idx.restype = f64
    L.orc_matsum_idx.argtypes = [i64, _dp, _ip, i64, _ip, i64]
    L.orc_vecsum_idx.restype = f64
    L.orc_vecsum_idx.argtypes = [_dp, _ip, i64]
    _bp = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    L.orc_pairwise_euclidean.argtypes = [i64, i64, _dp, _dp]
    L.orc_uniform_mh.restype = f64
    L.orc_uniform_mh.argtypes = [u64, u64, u64, u64]
    L.orc_mh_proposal.argtypes = [i64, _dp, _dp, C.c_void_p, C.c_void_p, i32, i32, _ip, _ip, C.POINTER(i64), PP, f64, f64,
                                  i64, u64, u64, u64, i32, C.POINTER(OrcMHInfo)]
    L.orc_sample_labels.argtypes = [i64, _dp, _dp, C.c_void_p, C.c_void_p, i32, i32, C.c_void_p, _ip, _ip,
                                    C.POINTER(i64), PP, f64, f64, i64, i64, u64, u64, i32, _bp, _bp]
    L.orc_sample_r.restype = f64
    L.orc_sample_r.argtypes = [u64, u64, f64, f64, _ip, i64, f64, f64, f64, C.POINTER(C.c_int)]
    L.orc_sample_p.restype = f64
    L.orc_sample_p.argtypes = [u64, u64, i64, i64, f64, f64, f64]
    L.orc_scalar_uniform.restype = f64
    L.orc_scalar_uniform.argtypes = [u64, u64, C.c_uint32, u64]
    L.orc_pair_measures_eval.restype = None
    L.orc_pair_measures_eval.argtypes = [i64, _ip, _ip, C.POINTER(OrcPairMeasures)]
    L.orc_mpel.restype = i64
    L.orc_mpel.argtypes = [i64, i64, _ip, i32, C.c_void_p, _dp]
    _lib = L
    return L


class OrcPairMeasures(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("ari", "ri", "mirkin", "hubert", "mi", "nmi", "vi", "ha", "hb", "id", "nid")]


def sample_r(seed, it, r, p, sizes, eta, sigma, proposalsd_r):
    """orc_sample_r: (new r, accepted).  sizes: non-empty cluster sizes in ascending label order."""
    Cs = np.ascontiguousarray(sizes, dtype=np.int64)
    acc = C.c_int()
    out = lib().orc_sample_r(seed, it, r, p, Cs, len(Cs), eta, sigma, proposalsd_r, C.byref(acc))
    return out, bool(acc.value)


def sample_p(seed, it, K, n, r, u, v):
    return lib().orc_sample_p(seed, it, K, n, r, u, v)


def run_chain(orc, init, r, p, numiters, burnin, thin, numGibbs, numMH, seed, proposalsd_r=1.0, rp_trace=None,
              stable=True, intended=False):
    """runsampler's loop (mcmc.jl:533-556) on an Oracle: sample_r!, sample_p!, sample_labels!, recording rule."""
    orc.set_state(init)
    P = orc.P
    rec = dict(r=[], p=[], K=[], loglik=[], logposterior=[], clusts=[], r_acc=[], sm_acc=[], sm_split=[], r_all=[], p_all=[])
    for i in range(1, numiters + 1):
        if rp_trace is None:
            sizes = orc.sizes[orc.sizes > 0]
            r, acc = sample_r(seed, i - 1, r, p, sizes, P.eta, P.sigma, proposalsd_r)
            p = sample_p(seed, i - 1, len(sizes), orc.n, r, P.u, P.v)
            rec["r_acc"].append(acc)
        else:
            r, p = float(rp_trace[0][i - 1]), float(rp_trace[1][i - 1])
        rec["r_all"].append(r); rec["p_all"].append(p)
        if numMH > 0 and not intended:
            _, a, s = orc.sample_labels(r, p, numMH, numGibbs, seed, i - 1, mode=1 if stable else 0)
            rec["sm_acc"] += list(a); rec["sm_split"] += list(s)
        else:
            if numMH > 0:
                for mh in range(numMH):
                    info = orc.mh_proposal(r, p, numGibbs, seed, i - 1, mh, mode=1 if stable else 0)
                    rec["sm_acc"].append(bool(info.accept)); rec["sm_split"].append(bool(info.split))
            (orc.sweep_stable if stable else orc.sweep_literal)(r, p, seed, i - 1)
        if i > burnin and (i - burnin) % thin == 0:
            ll = orc.loglik_stable() if stable else orc.loglik_literal()
            rec["clusts"].append(orc.sortlabels()); rec["K"].append(orc.K); rec["r"].append(r); rec["p"].append(p)
            rec["loglik"].append(ll); rec["logposterior"].append(ll + orc.logprior(r, p))
    return {k: np.array(v) for k, v in rec.items()}


def sweep_table(P, A, row_label, T_D, T_L, diag, eD, eL, clusts, r, p, seed, sweep, xs, XD, XL):
    """orc_sweep_table: one stable-arithmetic sweep driven by the row-sum table (rows = ascending non-empty labels,
    j = i included) and the fixed-point matrix rows XD / XL of the points xs that may change.
    Returns (labels, sizes, K, n_changes); raises AssertionError if a point outside xs changed."""
    n = len(clusts)
    c = np.ascontiguousarray(clusts, dtype=np.int64).copy()
    sizes = np.bincount(c, minlength=n + 1)[1:].astype(np.int64)
    K, ch = C.c_int64(), C.c_int64()
    xs = np.ascontiguousarray(xs, dtype=np.int64)
    XD = np.ascontiguousarray(XD, dtype=np.int64).reshape(-1) if len(xs) else np.zeros(1, np.int64)
    XL = np.ascontiguousarray(XL, dtype=np.int64).reshape(-1) if len(xs) else np.zeros(1, np.int64)
    rl = np.ascontiguousarray(row_label, dtype=np.int64)
    rc = lib().orc_sweep_table(n, len(rl), rl, np.ascontiguousarray(T_D, dtype=np.int64).reshape(-1),
                               np.ascontiguousarray(T_L, dtype=np.int64).reshape(-1),
                               np.ascontiguousarray(diag, dtype=np.int64), int(eD), int(eL),
                               np.ascontiguousarray(A, dtype=np.float64), c, sizes, C.byref(K), C.byref(params(P)),
                               float(r), float(p), int(seed), int(sweep), len(xs), xs if len(xs) else np.zeros(1, np.int64),
                               XD, XL, C.byref(ch))
    assert rc == 0, f"orc_sweep_table returned {rc} (-3: a point outside the supplied change set moved)"
    return c, sizes, K.value, ch.value


def size_table(P, n):
    A = np.zeros(n + 1)
    lib().orc_size_table(n, C.byref(params(P)), A)
    return A


def pair_measures(a, b) -> dict:
    a = np.ascontiguousarray(a, dtype=np.int64)
    b = np.ascontiguousarray(b, dtype=np.int64)
    out = OrcPairMeasures()
    lib().orc_pair_measures_eval(len(a), a, b, C.byref(out))
    return {k: getattr(out, k) for k, _ in OrcPairMeasures._fields_}


def mpel(samples, kind):
    """(0-based argmin, loss matrix, column sums) of orc_mpel"""
    S = np.ascontiguousarray(samples, dtype=np.int64)
    m, n = S.shape
    Lm = np.zeros((m, m))
    cs = np.zeros(m)
    i = lib().orc_mpel(m, n, S.reshape(-1), int(kind), Lm.ctypes.data_as(C.c_void_p), cs)
    return int(i), Lm, cs


def params(P: dict) -> OrcParams:
    return OrcParams(P["delta1"], P["delta2"], P["alpha"], P["beta"], P["zeta"], P["gamma"],
                     P.get("eta", 1.0), P.get("sigma", 1.0), P.get("u", 1.0), P.get("v", 1.0),
                     int(bool(P.get("repulsion", True))), int(P.get("maxK", 0)))


class Oracle:
    """Convenience wrapper holding one dataset (D, logD, fixed-point copies) and one label state."""

    def __init__(self, D: np.ndarray, P: dict, logD: np.ndarray | None = None, bits: int = 64, eL: int | None = None,
                 eD: int | None = None):
        self.L = lib()
        self.n = int(D.shape[0])
        self.D = np.ascontiguousarray(D, dtype=np.float64)
        if logD is None:
            logD = np.empty_like(self.D)
            if self.L.orc_make_logD(self.n, self.D, logD) != 0:
                raise ValueError("D must be symmetric.")
        self.logD = np.ascontiguousarray(logD, dtype=np.float64)
        self.set_params(P)
        nn = self.n * self.n
        if bits == 64:
            self.eD = self.L.orc_quant_exponent(self.n, self.D.ravel(), nn)
            self.eL = self.L.orc_quant_exponent(self.n, self.logD.ravel(), nn)
        else:
            self.eD = self.L.orc_quant_exponent32(self.D.ravel(), nn)
            self.eL = self.L.orc_quant_exponent32(self.logD.ravel(), nn)
        if eL is not None:   # the library's derived-logD mode caps the exponent (|logD·2^eL| < 2^51)
            self.eL = int(eL)
        if eD is not None:   # ... and D's (every Dq < 2^51, binding for n <= 1024): take the exponents the library reports
            self.eD = int(eD)
        self.Dq = np.empty((self.n, self.n), np.int64)
        self.Lq = np.empty((self.n, self.n), np.int64)
        self.L.orc_quantize(self.D.ravel(), nn, self.eD, self.Dq.reshape(-1))
        self.L.orc_quantize(self.logD.ravel(), nn, self.eL, self.Lq.reshape(-1))
        self.clusts = None
        self.sizes = None
        self.K = 0

    def set_params(self, P: dict):
        self.Pd = dict(P)
        self.P = params(P)
        self.A = np.zeros(self.n + 1)
        self.L.orc_size_table(self.n, C.byref(self.P), self.A)

    def set_state(self, clusts):
        self.clusts = np.ascontiguousarray(clusts, dtype=np.int64).copy()
        self.sizes = np.zeros(self.n, np.int64)
        K = C.c_int64()
        if self.L.orc_state_from_labels(self.n, self.clusts, self.sizes, C.byref(K)) != 0:
            raise ValueError("labels must lie in 1..n")
        self.K = K.value

    def sweep_literal(self, r, p, seed, sweep, cost_mode=0):
        K = C.c_int64()
        rc = self.L.orc_sweep_literal(self.n, self.D.reshape(-1), self.logD.reshape(-1), self.clusts, self.sizes,
                                      C.byref(K), C.byref(self.P), r, p, seed, sweep, cost_mode)
        assert rc == 0
        self.K = K.value
        return self.K

    def sweep_literal_range(self, r, p, seed, sweep, cost_mode, i_begin, i_end):
        K = C.c_int64()
        rc = self.L.orc_sweep_literal_range(self.n, self.D.reshape(-1), self.logD.reshape(-1), self.clusts,
                                            self.sizes, C.byref(K), C.byref(self.P), r, p, seed, sweep, cost_mode,
                                            i_begin, i_end)
        assert rc == 0
        self.K = K.value
        return self.K

    def mh_proposal(self, r, p, numGibbs, seed, it, mh, mode=0):
        """One split–merge proposal on the oracle's state (replaced by the final state on acceptance)."""
        info = OrcMHInfo()
        K = C.c_int64(self.K)
        dq = self.Dq.ctypes.data_as(C.c_void_p) if mode else None
        lq = self.Lq.ctypes.data_as(C.c_void_p) if mode else None
        self.L.orc_mh_proposal(self.n, self.D.reshape(-1), self.logD.reshape(-1), dq, lq, self.eD, self.eL, self.clusts,
                               self.sizes, C.byref(K), C.byref(self.P), r, p, numGibbs, seed, it, mh, mode, C.byref(info))
        self.K = K.value
        return info

    def sample_labels(self, r, p, numMH, numGibbs, seed, it, mode=0):
        """sample_labels! as written (numMH proposals + the Gibbs sweep, quirk Q1).  mode 0 literal, 1 stable."""
        acc = np.zeros(max(numMH, 1), np.uint8)
        spl = np.zeros(max(numMH, 1), np.uint8)
        K = C.c_int64(self.K)
        dq = self.Dq.ctypes.data_as(C.c_void_p) if mode else None
        lq = self.Lq.ctypes.data_as(C.c_void_p) if mode else None
        A = self.A.ctypes.data_as(C.c_void_p) if mode else None
        na = self.L.orc_sample_labels(self.n, self.D.reshape(-1), self.logD.reshape(-1), dq, lq, self.eD, self.eL, A,
                                      self.clusts, self.sizes, C.byref(K), C.byref(self.P), r, p, numMH, numGibbs, seed,
                                      it, mode, acc, spl)
        self.K = K.value
        return na, acc[:numMH].astype(bool), spl[:numMH].astype(bool)

    def sweep_stable(self, r, p, seed, sweep):
        K = C.c_int64()
        ch = C.c_int64()
        rc = self.L.orc_sweep_stable(self.n, self.Dq.reshape(-1), self.Lq.reshape(-1), self.eD, self.eL, self.A,
                                     self.clusts, self.sizes, C.byref(K), C.byref(self.P), r, p, seed, sweep,
                                     C.byref(ch))
        assert rc == 0
        self.K = K.value
        self.last_changes = ch.value
        return self.K

    def point_scores_literal(self, r, p, i):
        cands = np.zeros(self.n + 1, np.int64)
        lp = np.zeros(self.n + 1)
        m = self.L.orc_point_scores_literal(self.n, self.D.reshape(-1), self.logD.reshape(-1), self.clusts,
                                            self.sizes, C.byref(self.P), r, p, i, cands, lp)
        return cands[:m].copy(), lp[:m].copy()

    def point_scores_stable(self, r, p, i):
        cands = np.zeros(self.n + 1, np.int64)
        sc = np.zeros(self.n + 1)
        m = self.L.orc_point_scores_stable(self.n, self.Dq.reshape(-1), self.Lq.reshape(-1), self.eD, self.eL,
                                           self.A, self.clusts, self.sizes, C.byref(self.P), r, p, i, cands, sc)
        return cands[:m].copy(), sc[:m].copy()

    def loglik_literal(self):
        return self.L.orc_loglik_literal(self.n, self.D.reshape(-1), self.logD.reshape(-1), self.clusts,
                                         self.sizes, C.byref(self.P))

    def loglik_stable(self):
        return self.L.orc_loglik_stable(self.n, self.Dq.reshape(-1), self.Lq.reshape(-1), self.eD, self.eL,
                                        self.clusts, self.sizes, C.byref(self.P))

    def logprior(self, r, p):
        return self.L.orc_logprior(self.n, self.sizes, r, p, C.byref(self.P))

    def sortlabels(self):
        y = np.zeros(self.n, np.int64)
        self.L.orc_sortlabels(self.n, self.clusts, y)
        return y
