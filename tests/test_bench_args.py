"""bench.py's launch logic (CPU): `--gpus N` must really put N GPUs to work or fail loudly — a SCALE run that silently
measured one GPU and printed "n_gpus": 1 was the defect of round 2."""
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (importing bench does not import torch or touch a GPU)


def plan(gpus, **env):
    return bench.launch_plan(types.SimpleNamespace(gpus=gpus), env)


def test_launch_plan():
    assert "torch" not in bench.__dict__                       # nothing GPU-related is imported at module level
    assert plan(1) == "rank"
    assert plan(8) == "ranks"                                  # python bench.py --gpus 8: this process only launches the ranks
    assert plan(8, WORLD_SIZE="8", RANK="3", LOCAL_RANK="3") == "rank"   # the driver's torch.distributed.run form
    assert plan(1, WORLD_SIZE="1") == "rank"
    with pytest.raises(SystemExit) as e:
        plan(8, WORLD_SIZE="2")                                # launcher and flag disagree: refuse
    assert e.value.code == 2
    with pytest.raises(SystemExit):
        plan(1, WORLD_SIZE="4")
    assert plan(4, RC_BENCH_SINGLE_PROCESS="1") == "single"
    assert plan(4, RC_BENCH_SINGLE_PROCESS="0") == "ranks"
    with pytest.raises(SystemExit):
        plan(4, RC_BENCH_SINGLE_PROCESS="1", WORLD_SIZE="4")
    with pytest.raises(SystemExit):
        plan(0)


def test_gpus_2_without_two_gpus_fails_loudly():
    """No GPU in this container (one on the pool's boxes): `bench.py --gpus 2` starts two ranks in a fresh child, they
    cannot get two GPUs, and the run exits non-zero WITHOUT printing a result line — instead of measuring one GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RC_BENCH_SINGLE_PROCESS")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode != 0
    assert b'"metric"' not in r.stdout
    assert b"2-rank child exited" in r.stderr
    # the single-process form refuses too
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"],
                       env=dict(env, RC_BENCH_SINGLE_PROCESS="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode != 0 and b'"metric"' not in r.stdout


def test_median_and_windows_helpers():
    assert bench.median([3.0, 1.0, 2.0]) == 2.0 and bench.median([4.0, 1.0, 2.0, 3.0]) == 2.5

    class Fake:
        def __init__(self): self.calls = 0; self.syncs = 0
        def gibbs_sweep(self, r, p, seed, sweep, blocking=True): self.calls += 1
        def synchronize(self): self.syncs += 1
    f, bars = Fake(), []
    times, nxt = bench.timed_windows(f, 1, 1.0, 0.5, steps=7, warmup=3, windows=5, sync_all=lambda: bars.append(1), sweep0=10)
    assert len(times) == 5 and f.calls == 3 + 5 * 7 and nxt == 10 + 3 + 35 and len(bars) == 10   # exactly K steps per window, a barrier on both sides
