"""The logarithms k_resolve scores candidates with (rc_flog / rc_flog1p / rc_gumbel, csrc/redclust_hip.hip) against libm in
extended precision.  The reference evaluates log1p and log of Float64 (src/mcmc.jl:223-241, src/utils.jl:4); the kernel's own
table-driven routine must stay within 2 ulp of the true value on the domains the sweep feeds it (stated bound: 1.5 ulp)."""
import numpy as np
import pytest

import redclust_amd as rc

pytestmark = pytest.mark.gpu


def _ulps(got, ref_ld):
    ref = ref_ld.astype(np.float64)
    ulp = np.abs(np.nextafter(ref, np.inf) - ref)
    return np.abs((got.astype(np.longdouble) - ref_ld) / ulp.astype(np.longdouble)).astype(np.float64)


@pytest.fixture(scope="module")
def ctx():
    d = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "paper_datasets.npz"))
    c = rc.Context(d["D1"], device=0)
    yield c
    c.close()


def test_log_wide_range(ctx):
    rng = np.random.default_rng(11)
    x = np.exp((rng.random(2_000_000) - 0.5) * 160.0)
    x = np.concatenate([x, [1.0, 0.6875, 1.375, np.nextafter(1.0, 0), np.nextafter(1.0, 2), 2.0 ** -1000, 2.0 ** 1000]])
    got = ctx.debug_flog(0, x)
    err = _ulps(got, np.log(x.astype(np.longdouble)))
    assert got[len(x) - 7] == 0.0                      # log(1) = 0 exactly
    assert err.max() <= 2.0, (err.max(), x[err.argmax()])


def test_log_of_uniforms_keeps_relative_accuracy_near_one(ctx):
    rng = np.random.default_rng(12)
    k = rng.integers(1, 53, 1_000_000)
    u = np.concatenate([1.0 - np.ldexp(rng.random(1_000_000), -k), rng.random(1_000_000), [1.0 - 2.0 ** -53, 0.5 * 2.0 ** -52]])
    u = u[(u > 0) & (u < 1)]
    err = _ulps(ctx.debug_flog(0, u), np.log(u.astype(np.longdouble)))
    assert err.max() <= 2.0, (err.max(), u[err.argmax()])
    g = ctx.debug_flog(2, u)
    ref = -np.log(-np.log(u.astype(np.longdouble)))
    assert np.abs((g.astype(np.longdouble) - ref).astype(np.float64)).max() <= 2e-14      # absolute: the noise is added to scores of size >= 1
    assert g.max() <= 36.74                                                                # RC_GUMBEL_MAX of the pruning bound


def test_log1p_nonnegative(ctx):
    rng = np.random.default_rng(13)
    x = np.concatenate([np.exp((rng.random(2_000_000) - 0.67) * 60.0), [0.0, 2.0 ** -60, 1.0, 2.0 ** 40]])
    got = ctx.debug_flog(1, x)
    err = _ulps(got, np.log1p(x.astype(np.longdouble)))
    assert got[len(x) - 4] == 0.0
    assert err[np.isfinite(err)].max() <= 2.0, (err.max(), x[np.nanargmax(err)])
