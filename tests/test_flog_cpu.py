"""The table-driven logarithms of the sweep kernel (rc_flog / rc_flog1p / rc_gumbel), as a C replica on the CPU
(tests/flog_replica.c: the same operations in the same order) against long double libm.  The reference calls Julia's log1p / log
(src/mcmc.jl:223-241, src/utils.jl:4); DESIGN.md section 4 states <= 1.5 ulp for the kernel's routine on the sweep's domains —
asserted here without a GPU, and on the device itself by tests/test_gpu_logs.py."""
import os
import subprocess


def test_flog_replica_accuracy(tmp_path):
    src = os.path.join(os.path.dirname(__file__), "flog_replica.c")
    exe = str(tmp_path / "flog_replica")
    subprocess.check_call(["gcc", "-O2", "-o", exe, src, "-lm"])
    out = subprocess.check_output([exe, "1500000"], text=True).split()
    e_log, e_u, e_1p, e_g, top, log_one, log1p_zero = map(float, out)
    assert e_log <= 1.5 and e_u <= 1.5 and e_1p <= 1.6, (e_log, e_u, e_1p)      # ulp, wide range / u in (0, 1) incl. close to 1 / log1p(x >= 0)
    assert e_g <= 2e-14                                                          # the Gumbel noise, absolute
    assert 36.73 < top <= 36.74                                                  # RC_GUMBEL_MAX of the pruning bound covers the largest noise value
    assert log_one == 0.0 and log1p_zero == 0.0
