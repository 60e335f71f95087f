"""ctypes binding of libredclust_hip.so (include/redclust_hip.h).  No CPU fallback: if the HIP library is
missing or no GPU is present, every compute entry point raises."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
SO = os.environ.get("RC_LIB_PATH") or os.path.join(CSRC, "libredclust_hip.so")   # RC_LIB_PATH: experiment builds
HEADER = os.path.join(ROOT, "include", "redclust_hip.h")

RC_OK = 0
ERRORS = {-1: "RC_ERR_ARG", -2: "RC_ERR_HIP", -3: "RC_ERR_OOM", -4: "RC_ERR_DOMAIN", -5: "RC_ERR_STATE",
          -6: "RC_ERR_CAPACITY"}


class RedClustHIPError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERRORS.get(code, code)}: {msg}")
        self.code = code


class RedClustDomainError(RedClustHIPError, ValueError):
    """RC_ERR_DOMAIN: the input violates what MCMCData requires (src/types.jl:145-157) — the reference throws
    ArgumentError there, so this one is a ValueError as well."""


def _error(code, msg):
    return (RedClustDomainError if code == -4 else RedClustHIPError)(code, msg)


class RcParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("delta1", "delta2", "alpha", "beta", "zeta", "gamma", "eta", "sigma",
                                          "u", "v")] + [("maxK", C.c_int64), ("repulsion", C.c_uint8),
                                                        ("pad_", C.c_uint8 * 7)]


class RcPairMeasures(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("ari", "ri", "mirkin", "hubert", "mi", "nmi", "vi", "ha", "hb", "id", "nid")]


class RcWbStats(C.Structure):
    _fields_ = [("count_within", C.c_int64), ("count_between", C.c_int64), ("sum_within", C.c_double),
                ("sumlog_within", C.c_double), ("sum_between", C.c_double), ("sumlog_between", C.c_double)]


class RcChainOptions(C.Structure):
    _fields_ = [("numiters", C.c_int64), ("burnin", C.c_int64), ("thin", C.c_int64), ("numGibbs", C.c_int64),
                ("numMH", C.c_int64), ("splitmerge_mode", C.c_int32), ("pad_", C.c_int32), ("seed", C.c_uint64),
                ("first_iter", C.c_uint64), ("r0", C.c_double), ("p0", C.c_double), ("proposalsd_r", C.c_double),
                ("r_trace", C.c_void_p), ("p_trace", C.c_void_p), ("max_samples", C.c_int64)]


class RcChainOutputs(C.Structure):
    _fields_ = [("clusts", C.c_void_p), ("K", C.c_void_p), ("r", C.c_void_p), ("p", C.c_void_p), ("loglik", C.c_void_p),
                ("logposterior", C.c_void_p), ("r_acceptances", C.c_void_p), ("splitmerge_acceptances", C.c_void_p),
                ("splitmerge_splits", C.c_void_p), ("r_all", C.c_void_p), ("p_all", C.c_void_p),
                ("num_samples", C.c_int64), ("runtime_s", C.c_double), ("r_final", C.c_double), ("p_final", C.c_double)]


class RcChainsInput(C.Structure):
    _fields_ = [("n", C.c_int64), ("D", C.c_void_p), ("logD_or_null", C.c_void_p), ("points", C.c_void_p), ("dim", C.c_int64),
                ("storage_bits", C.c_int32), ("pad_", C.c_int32), ("kcap", C.c_int64), ("params", C.POINTER(RcParams)),
                ("init_clusts", C.c_void_p)]


class RcSweepStats(C.Structure):
    _fields_ = [("n_changes", C.c_int64), ("n_rounds", C.c_int64), ("K", C.c_int64)]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, "redclust_hip.hip"), os.path.join(CSRC, "pointestimate.inc.hip"),
            os.path.join(CSRC, "chain.inc.hip"), os.path.join(CSRC, "chains.inc.hip"), HEADER]
    if not force and os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(s) for s in srcs):
        return SO
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", SO, srcs[0]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return SO


def build_diag(extra_defines=(), name="libredclust_hip_diag.so", verbose: bool = False) -> str:
    """The tuning / diagnostic build (-DRC_DIAG: the RC_* environment switches INTEGRATION.md lists as diagnostic exist only
    here; extra_defines e.g. ("RC_CHAOS=15",), ("RC_POISON",)).  Written to build_exp/ (git- and gpurun-ignored: build it on
    the box that uses it, select it with RC_LIB_PATH)."""
    out_dir = os.path.join(ROOT, "build_exp")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, name)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DRC_DIAG"] + [f"-D{d}" for d in extra_defines] + \
          ["-o", so, os.path.join(CSRC, "redclust_hip.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return so


_dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_up = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")

# every symbol include/redclust_hip.h declares, with its ctypes signature
SIGNATURES = {
    "rc_create": (C.c_int32, [C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_void_p)]),
    "rc_create_from_points": (C.c_int32, [C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_void_p)]),
    "rc_get_matrix": (C.c_int32, [C.c_void_p, C.c_int32, _dp]),
    "rc_get_matrix_rows": (C.c_int32, [C.c_void_p, C.c_int32, _ip, C.c_int64, _dp]),
    "rc_destroy": (C.c_int32, [C.c_void_p]),
    "rc_last_error": (C.c_char_p, [C.c_void_p]),
    "rc_set_params": (C.c_int32, [C.c_void_p, C.POINTER(RcParams)]),
    "rc_set_state": (C.c_int32, [C.c_void_p, _ip]),
    "rc_get_state": (C.c_int32, [C.c_void_p, C.c_void_p, _ip, C.POINTER(C.c_int64)]),
    "rc_gibbs_sweep": (C.c_int32, [C.c_void_p, C.c_double, C.c_double, C.c_uint64, C.c_uint64]),
    "rc_gibbs_sweep_async": (C.c_int32, [C.c_void_p, C.c_double, C.c_double, C.c_uint64, C.c_uint64]),
    "rc_last_sweep_stats": (C.c_int32, [C.c_void_p, C.POINTER(RcSweepStats)]),
    "rc_synchronize": (C.c_int32, [C.c_void_p]),
    "rc_loglik": (C.c_int32, [C.c_void_p, C.POINTER(C.c_double)]),
    "rc_logprior": (C.c_int32, [C.c_void_p, C.c_double, C.c_double, C.POINTER(C.c_double)]),
    "rc_record_sample": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "rc_cocluster": (C.c_int32, [C.c_void_p, _dp, C.c_int64]),
    "rc_cocluster_counts": (C.c_int32, [C.c_void_p, _up]),
    "rc_cocluster_device_buffer": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "rc_cocluster_reset": (C.c_int32, [C.c_void_p]),
    "rc_set_mode": (C.c_int32, [C.c_void_p, C.c_int32]),
    "rc_attach_host_matrices": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rc_splitmerge": (C.c_int32, [C.c_void_p, C.c_double, C.c_double, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint64,
                                  C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]),
    "rc_state_checkpoint": (C.c_int32, [C.c_void_p]),
    "rc_state_restore": (C.c_int32, [C.c_void_p]),
    "rc_debug_rowsums": (C.c_int32, [C.c_void_p, C.c_int64, _ip, _ip, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rc_debug_rowtotals": (C.c_int32, [C.c_void_p, _ip, _ip]),
    "rc_debug_flog": (C.c_int32, [C.c_void_p, C.c_int32, _dp, C.c_int64, _dp]),
    "rc_bulk_kernel_info": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "rc_set_bulk_kernel": (C.c_int32, [C.c_void_p, C.c_int32]),
    "rc_set_option": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_int64]),
    "rc_bulk_kernel_name": (C.c_char_p, [C.c_void_p]),
    "rc_within_between": (C.c_int32, [C.c_void_p, C.POINTER(RcWbStats)]),
    "rc_run_chain": (C.c_int32, [C.c_void_p, C.POINTER(RcChainOptions), C.POINTER(RcChainOutputs)]),
    "rc_comm_unique_id": (C.c_int32, [C.c_void_p]),
    "rc_comm_create": (C.c_int32, [C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "rc_comm_destroy": (C.c_int32, [C.c_void_p]),
    "rc_comm_allreduce_counts": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                             C.POINTER(C.c_double)]),
    "rc_run_chains": (C.c_int32, [C.c_int32, C.POINTER(C.c_int32), C.POINTER(RcChainsInput), C.POINTER(RcChainOptions),
                                  C.POINTER(RcChainOutputs), C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "rc_measure_read_ceiling": (C.c_int32, [C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_double)]),
    "rc_scalar_updates": (C.c_int32, [C.c_uint64, C.c_uint64, C.c_double, C.c_double, _ip, C.c_int64, C.c_int64, C.c_double,
                                      C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.POINTER(C.c_uint8)]),
    "rc_loss_matrix": (C.c_int32, [C.c_int32, _ip, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "rc_pair_measures": (C.c_int32, [C.c_int32, _ip, _ip, C.c_int64, C.POINTER(RcPairMeasures)]),
    "rc_layout_info": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rc_event_overhead_ms": (C.c_int32, [C.c_void_p, C.POINTER(C.c_double)]),
    "rc_kernel_timing": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "rc_capacity_info": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "rc_chain_stats": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
}

_lib = None


def lib():
    """Load the in-tree HIP library; fails loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise RedClustHIPError(-2, f"{SO} not built — run `python -c 'import __graft_entry__ as g; g.build()'` "
                                       "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(SO)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("RC_LIB_PATH") and not hasattr(L, name):
                continue   # experiment builds of older sources lack the newest entry points
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class Context:
    """Owns one rc_ctx: device-resident D/logD (fixed point), the label state and the sweep kernels."""

    def __init__(self, D: np.ndarray, logD: np.ndarray | None = None, device: int = 0, kcap: int = 0,
                 storage_bits: int = 64):
        self.L = lib()
        D = np.ascontiguousarray(D, dtype=np.float64)
        if D.ndim != 2 or D.shape[0] != D.shape[1]:
            raise ValueError("D must be a square matrix.")  # types.jl:152-154
        self.n = int(D.shape[0])
        lp = None
        if logD is not None:
            logD = np.ascontiguousarray(logD, dtype=np.float64)
            assert logD.shape == D.shape
            lp = logD.ctypes.data_as(C.c_void_p)
        h = C.c_void_p()
        rc = self.L.rc_create(self.n, D.ctypes.data_as(C.c_void_p), lp, storage_bits, device, kcap, C.byref(h))
        if rc != RC_OK:
            raise _error(rc, self.L.rc_last_error(None).decode())
        self.h = h

    @classmethod
    def from_points(cls, points, device: int = 0, kcap: int = 0, storage_bits: int = 64):
        """MCMCData(points) on the device: pairwise Euclidean distances are computed there (types.jl:159-162)."""
        self = cls.__new__(cls)
        self.L = lib()
        pts = np.ascontiguousarray(points, dtype=np.float64)
        if pts.ndim != 2:
            raise ValueError("points must be an n×dim array (one observation per row)")
        self.n = int(pts.shape[0])
        h = C.c_void_p()
        rc = self.L.rc_create_from_points(self.n, int(pts.shape[1]), pts.ctypes.data_as(C.c_void_p), storage_bits, device,
                                          kcap, C.byref(h))
        if rc != RC_OK:
            raise _error(rc, self.L.rc_last_error(None).decode())
        self.h = h
        return self

    def get_matrix(self, which=0):
        out = np.zeros((self.n, self.n))
        self._chk(self.L.rc_get_matrix(self.h, int(which), out.reshape(-1)))
        return out

    def get_matrix_rows(self, which, rows):
        """rows (0-based, caller's order) of the device's D (which=0) or logD (which=1) as an len(rows)×n array"""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        out = np.zeros((len(rows), self.n))
        if len(rows):
            self._chk(self.L.rc_get_matrix_rows(self.h, int(which), rows, len(rows), out.reshape(-1)))
        return out

    def _chk(self, rc):
        if rc != RC_OK:
            raise _error(rc, self.L.rc_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.rc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, delta1, delta2, alpha, beta, zeta, gamma, eta=1.0, sigma=1.0, u=1.0, v=1.0,
                   repulsion=True, maxK=0, **_ignored):
        P = RcParams(delta1, delta2, alpha, beta, zeta, gamma, eta, sigma, u, v, int(maxK), int(bool(repulsion)))
        self._chk(self.L.rc_set_params(self.h, C.byref(P)))

    def set_state(self, clusts):
        c = np.ascontiguousarray(clusts, dtype=np.int64)
        if c.shape != (self.n,):
            raise ValueError("clusts must have length n")
        self._chk(self.L.rc_set_state(self.h, c))

    def get_state(self, want_labels=True):
        """(clusts, clustsizes, K); with want_labels=False clusts is None and nothing is copied from the device."""
        clusts = np.zeros(self.n, np.int64) if want_labels else None
        sizes = np.zeros(self.n, np.int64)
        K = C.c_int64()
        self._chk(self.L.rc_get_state(self.h, clusts.ctypes.data_as(C.c_void_p) if want_labels else None, sizes,
                                      C.byref(K)))
        return clusts, sizes, K.value

    def gibbs_sweep(self, r, p, seed, sweep_index, blocking=True):
        fn = self.L.rc_gibbs_sweep if blocking else self.L.rc_gibbs_sweep_async
        self._chk(fn(self.h, float(r), float(p), int(seed), int(sweep_index)))

    def set_mode(self, mode):
        """'full' (recompute the row-sum table every sweep, the reference's data flow) or 'incremental' (maintain it
        by exact corrections only); results are bit-identical."""
        self._chk(self.L.rc_set_mode(self.h, {"full": 0, "incremental": 1}[mode]))

    def synchronize(self):
        self._chk(self.L.rc_synchronize(self.h))

    def sweep_stats(self):
        s = RcSweepStats()
        self._chk(self.L.rc_last_sweep_stats(self.h, C.byref(s)))
        return dict(n_changes=s.n_changes, n_rounds=s.n_rounds, K=s.K)

    def loglik(self):
        out = C.c_double()
        self._chk(self.L.rc_loglik(self.h, C.byref(out)))
        return out.value

    def logprior(self, r, p):
        out = C.c_double()
        self._chk(self.L.rc_logprior(self.h, float(r), float(p), C.byref(out)))
        return out.value

    def record_sample(self, want_labels=True):
        if want_labels:
            out = np.zeros(self.n, np.int64)
            self._chk(self.L.rc_record_sample(self.h, out.ctypes.data_as(C.c_void_p)))
            return out
        self._chk(self.L.rc_record_sample(self.h, None))
        return None

    def cocluster(self, numsamples):
        out = np.zeros((self.n, self.n))
        self._chk(self.L.rc_cocluster(self.h, out.reshape(-1), int(numsamples)))
        return out

    def cocluster_counts(self):
        out = np.zeros((self.n, self.n), np.uint32)
        self._chk(self.L.rc_cocluster_counts(self.h, out.reshape(-1)))
        return out

    def cocluster_device_buffer(self):
        p = C.c_void_p()
        ld = C.c_int64()
        self._chk(self.L.rc_cocluster_device_buffer(self.h, C.byref(p), C.byref(ld)))
        return p.value, ld.value

    def cocluster_reset(self):
        self._chk(self.L.rc_cocluster_reset(self.h))

    def attach_host_matrices(self, D, logD=None):
        """Borrow the caller's host matrices for the split–merge scans; they must outlive the context's use."""
        self._hostD = np.ascontiguousarray(D, dtype=np.float64)
        self._hostL = None if logD is None else np.ascontiguousarray(logD, dtype=np.float64)
        self._chk(self.L.rc_attach_host_matrices(self.h, self._hostD.ctypes.data_as(C.c_void_p),
                                                 None if self._hostL is None else self._hostL.ctypes.data_as(C.c_void_p)))

    def splitmerge(self, r, p, numGibbs, seed, it, mh_counter):
        a, s = C.c_uint8(), C.c_uint8()
        self._chk(self.L.rc_splitmerge(self.h, float(r), float(p), int(numGibbs), int(seed), int(it), int(mh_counter),
                                       C.byref(a), C.byref(s)))
        return bool(a.value), bool(s.value)

    def checkpoint(self):
        self._chk(self.L.rc_state_checkpoint(self.h))

    def restore(self):
        self._chk(self.L.rc_state_restore(self.h))

    def debug_rowsums(self, label):
        sd = np.zeros(self.n, np.int64)
        sl = np.zeros(self.n, np.int64)
        eD, eL = C.c_int32(), C.c_int32()
        self._chk(self.L.rc_debug_rowsums(self.h, int(label), sd, sl, C.byref(eD), C.byref(eL)))
        return sd, sl, eD.value, eL.value

    def debug_rowtotals(self):
        """rc_debug_rowtotals: (Σ_j Dq[i,j], Σ_j Lq[i,j]) of every row, fixed point, by a kernel independent of the row reductions."""
        td = np.zeros(self.n, np.int64)
        tl = np.zeros(self.n, np.int64)
        self._chk(self.L.rc_debug_rowtotals(self.h, td, tl))
        return td, tl

    def debug_flog(self, which, x):
        """rc_debug_flog: the sweep kernel's own log (which = 0), log1p (1) or -log(-log(x)) (2) of every entry of x, on the device."""
        x = np.ascontiguousarray(x, np.float64)
        out = np.zeros_like(x)
        self._chk(self.L.rc_debug_flog(self.h, int(which), x, x.size, out))
        return out

    def bulk_kernel_info(self):
        w, b = C.c_int32(), C.c_double()
        self._chk(self.L.rc_bulk_kernel_info(self.h, C.byref(w), C.byref(b)))
        return ("k_bulk_sym" if w.value else "k_bulk"), b.value

    def bulk_kernel_name(self) -> str:
        return self.L.rc_bulk_kernel_name(self.h).decode()

    def set_bulk_kernel(self, which):
        self._chk(self.L.rc_set_bulk_kernel(self.h, {"auto": -1, "perm": 0, "sym": 1}[which]))

    def set_option(self, name, value):
        """rc_set_option: "prune" (-1 automatic / 0 / 1), "lds_point_cache" (0 / 1), "chain_workers", "chain_depth" (0 = automatic), "chain_pipeline" (0 / 1)."""
        self._chk(self.L.rc_set_option(self.h, name.encode(), int(value)))

    def run_chain(self, numiters, burnin, thin, numGibbs, numMH, seed, r0, p0, proposalsd_r, splitmerge="as_written",
                  rp_trace=None, first_iter=0):
        """rc_run_chain: the whole iteration loop natively.  Returns a dict of numpy arrays (MCMCResult fields)."""
        ns = max((numiters - burnin) // thin, 0) if thin > 0 else 0   # thin <= 0 is rejected by the library
        n = self.n
        res = dict(clusts=np.zeros((ns, n), np.int64), K=np.zeros(ns, np.int64), r=np.zeros(ns), p=np.zeros(ns),
                   loglik=np.zeros(ns), logposterior=np.zeros(ns), r_acceptances=np.zeros(numiters, np.uint8),
                   splitmerge_acceptances=np.zeros(numiters * numMH, np.uint8),
                   splitmerge_splits=np.zeros(numiters * numMH, np.uint8), r_all=np.zeros(numiters), p_all=np.zeros(numiters))
        o = RcChainOptions(numiters, burnin, thin, numGibbs, numMH, {"as_written": 0, "intended": 1}[splitmerge], 0,
                           int(seed), int(first_iter), float(r0), float(p0), float(proposalsd_r), None, None, ns)
        keep = []
        if rp_trace is not None:
            rt = np.ascontiguousarray(rp_trace[0], dtype=np.float64)
            pt = np.ascontiguousarray(rp_trace[1], dtype=np.float64)
            assert len(rt) >= numiters and len(pt) >= numiters
            keep = [rt, pt]
            o.r_trace, o.p_trace = rt.ctypes.data, pt.ctypes.data
        out = RcChainOutputs()
        for k in ("clusts", "K", "r", "p", "loglik", "logposterior", "r_acceptances", "splitmerge_acceptances",
                  "splitmerge_splits", "r_all", "p_all"):
            setattr(out, k, res[k].ctypes.data if res[k].size else None)
        self._chk(self.L.rc_run_chain(self.h, C.byref(o), C.byref(out)))
        del keep
        res["num_samples"], res["runtime_s"] = int(out.num_samples), float(out.runtime_s)
        res["r_final"], res["p_final"] = float(out.r_final), float(out.p_final)
        for k in ("r_acceptances", "splitmerge_acceptances", "splitmerge_splits"):
            res[k] = res[k].astype(bool)
        return res

    def capacity_info(self) -> dict:
        """Slot capacity now, its ceiling min(n, 32767) (wide beyond 4096 slots), growths so far, the resolver's batch capacity."""
        v = [C.c_int64(0) for _ in range(4)]
        self._chk(self.L.rc_capacity_info(self.h, *[C.byref(x) for x in v]))
        return dict(kcap=int(v[0].value), kcap_max=int(v[1].value), n_grows=int(v[2].value), batch_capacity=int(v[3].value))

    def chain_stats(self) -> dict:
        """Counters of the last run_chain: rollbacks of the speculative pipeline, off-line split evaluations, worker threads,
        capacity growths."""
        v = [C.c_int64(0) for _ in range(4)]
        self._chk(self.L.rc_chain_stats(self.h, *[C.byref(x) for x in v]))
        return dict(rollbacks=int(v[0].value), split_evals=int(v[1].value), workers=int(v[2].value), grows=int(v[3].value))

    def within_between(self) -> dict:
        """rc_within_between: |A|, ΣA, Σlog A and |B|, ΣB, Σlog B of fitprior's split under the current labels."""
        o = RcWbStats()
        self._chk(self.L.rc_within_between(self.h, C.byref(o)))
        return {k: getattr(o, k) for k, _ in RcWbStats._fields_}

    def layout_info(self):
        """(layouts built so far, label runs in the internal point order)"""
        a, b = C.c_int32(), C.c_int32()
        self._chk(self.L.rc_layout_info(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def event_overhead_ms(self):
        out = C.c_double()
        self._chk(self.L.rc_event_overhead_ms(self.h, C.byref(out)))
        return out.value

    def kernel_timing(self, enable=-1):
        ms = C.c_double()
        cnt = C.c_int64()
        self._chk(self.L.rc_kernel_timing(self.h, int(enable), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value


def _chain_buffers(n, numiters, burnin, thin, numMH):
    ns = max((numiters - burnin) // thin, 0) if thin > 0 else 0
    res = dict(clusts=np.zeros((ns, n), np.int64), K=np.zeros(ns, np.int64), r=np.zeros(ns), p=np.zeros(ns),
               loglik=np.zeros(ns), logposterior=np.zeros(ns), r_acceptances=np.zeros(numiters, np.uint8),
               splitmerge_acceptances=np.zeros(numiters * numMH, np.uint8),
               splitmerge_splits=np.zeros(numiters * numMH, np.uint8), r_all=np.zeros(numiters), p_all=np.zeros(numiters))
    out = RcChainOutputs()
    for k in res:
        setattr(out, k, res[k].ctypes.data if res[k].size else None)
    return ns, res, out


class Comm:
    """rc_comm: the RCCL communicator of the chains (one per GPU).  In one process: Comm(device_ids).  One process per
    GPU: rank 0 calls Comm.unique_id(), moves the bytes to the others, every rank calls Comm([device], rank, world, id)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        rc = lib().rc_comm_unique_id(buf)
        if rc != RC_OK:
            raise _error(rc, lib().rc_last_error(None).decode())
        return bytes(buf)

    def __init__(self, device_ids, rank_offset: int = 0, world_size: int | None = None, unique_id: bytes | None = None):
        self.L = lib()
        self.devices = [int(d) for d in device_ids]
        world = len(self.devices) if world_size is None else int(world_size)
        devs = (C.c_int32 * len(self.devices))(*self.devices)
        idb = (C.c_uint8 * 128)(*unique_id) if unique_id is not None else None
        h = C.c_void_p()
        rc = self.L.rc_comm_create(len(self.devices), devs, int(rank_offset), world, idb, C.byref(h))
        if rc != RC_OK:
            raise _error(rc, self.L.rc_last_error(None).decode())
        self.h = h

    def allreduce_counts(self, ctxs, num_samples):
        """In-place merge of the contexts' co-clustering counts over all chains.  Returns (total_samples, elapsed_ms)."""
        hs = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        ns = (C.c_int64 * len(ctxs))(*[int(x) for x in num_samples])
        tot, ms = C.c_int64(0), C.c_double(0)
        rc = self.L.rc_comm_allreduce_counts(self.h, hs, ns, C.byref(tot), C.byref(ms))
        if rc != RC_OK:
            raise _error(rc, self.L.rc_last_error(None).decode())
        return int(tot.value), float(ms.value)

    def close(self):
        if getattr(self, "h", None):
            self.L.rc_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_chains(device_ids, params: dict, init_clusts, numiters, burnin, thin, numGibbs, numMH, seed, r0, p0, proposalsd_r, *,
               D=None, logD=None, points=None, storage_bits: int = 64, kcap: int = 0, splitmerge="as_written",
               want_posterior: bool = True):
    """rc_run_chains: len(device_ids) chains in this process (one host thread + context per GPU, chain c seeded seed + c),
    merged over RCCL.  Returns (list of per-chain dicts as Context.run_chain, merged posterior co-clustering or None,
    total number of samples, all-reduce milliseconds)."""
    L = lib()
    devs = [int(d) for d in device_ids]
    nch = len(devs)
    inp = RcChainsInput()
    keep = []
    if D is not None:
        D = np.ascontiguousarray(D, dtype=np.float64); keep.append(D)
        n = int(D.shape[0]); inp.D = D.ctypes.data
        if logD is not None:
            logD = np.ascontiguousarray(logD, dtype=np.float64); keep.append(logD); inp.logD_or_null = logD.ctypes.data
    else:
        points = np.ascontiguousarray(points, dtype=np.float64); keep.append(points)
        n = int(points.shape[0]); inp.points = points.ctypes.data; inp.dim = int(points.shape[1])
    g = params.get
    prm = RcParams(g("delta1"), g("delta2"), g("alpha"), g("beta"), g("zeta"), g("gamma"), g("eta", 1.0), g("sigma", 1.0),
                   g("u", 1.0), g("v", 1.0), int(g("maxK", 0)), int(bool(g("repulsion", True))))
    init = np.ascontiguousarray(init_clusts, dtype=np.int64); keep.append(init)
    inp.n, inp.storage_bits, inp.kcap, inp.params, inp.init_clusts = n, int(storage_bits), int(kcap), C.pointer(prm), init.ctypes.data
    ns = 0
    ress, outs = [], (RcChainOutputs * nch)()
    for c in range(nch):
        ns, res, out = _chain_buffers(n, numiters, burnin, thin, numMH)
        ress.append(res); outs[c] = out
    o = RcChainOptions(numiters, burnin, thin, numGibbs, numMH, {"as_written": 0, "intended": 1}[splitmerge], 0,
                       int(seed), 0, float(r0), float(p0), float(proposalsd_r), None, None, ns)
    post = np.zeros((n, n)) if want_posterior else None
    tot, ms = C.c_int64(0), C.c_double(0)
    rc = L.rc_run_chains(nch, (C.c_int32 * nch)(*devs), C.byref(inp), C.byref(o), outs,
                         post.ctypes.data if post is not None else None, C.byref(tot), C.byref(ms))
    if rc != RC_OK:
        raise _error(rc, L.rc_last_error(None).decode())
    for c, res in enumerate(ress):
        res["num_samples"], res["runtime_s"] = int(outs[c].num_samples), float(outs[c].runtime_s)
        res["r_final"], res["p_final"] = float(outs[c].r_final), float(outs[c].p_final)
        for k in ("r_acceptances", "splitmerge_acceptances", "splitmerge_splits"):
            res[k] = res[k].astype(bool)
    del keep
    return ress, post, int(tot.value), float(ms.value)


def measure_read_ceiling(device: int = 0, mib: int = 2048, reps: int = 5) -> float:
    """rc_measure_read_ceiling: the device's streaming-read rate in GB/s as measured now (buffer >> Infinity Cache)."""
    L = lib()
    out = C.c_double(0)
    rc = L.rc_measure_read_ceiling(int(device), int(mib), int(reps), C.byref(out))
    if rc != RC_OK:
        raise _error(rc, L.rc_last_error(None).decode())
    return float(out.value)


def loss_matrix(samples, loss: int, device: int = 0, want_matrix: bool = True):
    """rc_loss_matrix: (lossmatrix or None, column sums, 0-based argmin, kernel ms) for an m×n int64 label matrix."""
    L = lib()
    S = np.ascontiguousarray(samples, dtype=np.int64)
    if S.ndim != 2:
        raise ValueError("samples must be an m×n matrix of labels")
    m, n = S.shape
    M = np.zeros((m, m)) if want_matrix else None
    cs = np.zeros(m)
    am, ms = C.c_int64(), C.c_double()
    rc = L.rc_loss_matrix(device, S.reshape(-1), m, n, int(loss), None if M is None else M.ctypes.data_as(C.c_void_p),
                          cs.ctypes.data_as(C.c_void_p), C.byref(am), C.byref(ms))
    if rc != RC_OK:
        raise RedClustHIPError(rc, L.rc_last_error(None).decode())
    return M, cs, int(am.value), float(ms.value)


def pair_measures(a, b, device: int = 0) -> dict:
    """rc_pair_measures as a dict."""
    L = lib()
    a = np.ascontiguousarray(a, dtype=np.int64)
    b = np.ascontiguousarray(b, dtype=np.int64)
    out = RcPairMeasures()
    rc = L.rc_pair_measures(device, a, b, len(a), C.byref(out))
    if rc != RC_OK:
        raise RedClustHIPError(rc, L.rc_last_error(None).decode())
    return {k: getattr(out, k) for k, _ in RcPairMeasures._fields_}
