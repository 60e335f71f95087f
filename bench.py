#!/usr/bin/env python3
"""bench.py — Gibbs sweeps/sec of the MI355X sweep on BASELINE.json's roofline configuration.

  python bench.py --gpus N --steps K --warmup W          (N=1: plain python; N>1: one rank per GPU under
  python -m torch.distributed.run --nproc-per-node N …    torch.distributed, backend nccl = RCCL)

A "step" is one Gibbs sweep (sample_labels_Gibbs!, /root/reference/src/mcmc.jl:158-256) of one chain over the
synthetic N=8192, K=50 dense Float64 dissimilarity matrix (BASELINE.json configs[2], the configuration the
north_star's roofline target is quoted on); D and logD are already resident in HBM when the timed region
starts.  Chains are independent: with N GPUs every rank runs its own chain (weak scaling) and the only
collective is the final sum all-reduce of the n×n co-clustering counts (outside the timed region, reported).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (k_bulk, HBM-bound, timed
with HIP events on the library's own stream) and `cpu_baseline` (the C restatement of the reference's loop,
faithful-cost mode, one host core, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_POINTS = int(os.environ.get("RC_BENCH_N", 8192))
N_CLUST = int(os.environ.get("RC_BENCH_K", 50))
BITS = int(os.environ.get("RC_BENCH_BITS", 64))  # 32: int32 fixed-point storage (BASELINE config 5 style)
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ≈6300 GB/s is the measured copy ceiling


def cpu_baseline(D, P, labels, r, p, max_seconds=20.0):
    """Reference loop restated in C (oracle/, literal arithmetic, faithful-cost: per-(point,cluster) member
    scans and the three strided gathers of mcmc.jl:195-214), single thread, on a bounded sample of one sweep."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    n = D.shape[0]
    orc = O.Oracle.__new__(O.Oracle)  # skip the fixed-point copies (1 GiB) — literal mode does not use them
    orc.L = O.lib()
    orc.n = n
    orc.D = np.ascontiguousarray(D)
    logD = np.log(np.where(np.eye(n, dtype=bool), 1.0, D))
    orc.logD = np.ascontiguousarray(logD)
    orc.P = O.params(P)
    orc.set_state(labels)
    pts = 16
    t0 = time.perf_counter()
    orc.sweep_literal_range(r, p, 1, 0, 1, 0, pts)
    dt = time.perf_counter() - t0
    # grow the sample to ≈ max_seconds of CPU work (capped at one full sweep)
    want = int(min(n, max(pts, pts * max_seconds / max(dt, 1e-9) * 0.8)))
    orc.set_state(labels)
    t0 = time.perf_counter()
    orc.sweep_literal_range(r, p, 1, 0, 1, 0, want)
    dt = time.perf_counter() - t0
    sweeps_per_s = 1.0 / (dt * n / want)
    # the same restatement with each row bucketed once ("single-pass"): what a tidy single-threaded CPU code would do
    orc.set_state(labels)
    t0 = time.perf_counter()
    orc.sweep_literal_range(r, p, 1, 0, 0, 0, n)
    dt_sp = time.perf_counter() - t0
    import shutil
    return {"value": sweeps_per_s, "unit": "sweeps/s", "cores": 1, "kind": "port",
            "single_pass_variant_sweeps_per_s": 1.0 / dt_sp,
            # time the Julia package itself if the box has it — it does not (probed, nothing installed)
            "julia_on_this_box": shutil.which("julia") is not None,
            "sample": f"first {want} of {n} points of one sweep (faithful-cost literal C restatement of "
                      f"mcmc.jl:158-256, single thread, {dt:.1f} s), scaled to a full sweep; host has {os.cpu_count()} cores"}


import contextlib


@contextlib.contextmanager
def stdout_to_stderr():
    """fd 1 points at stderr inside the block (C-level prints of libraries included); stdout is restored afterwards."""
    import ctypes
    libc = ctypes.CDLL(None)
    sys.stdout.flush(); libc.fflush(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush(); libc.fflush(None)
        os.dup2(saved, 1); os.close(saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    distributed = world > 1 or bool(os.environ.get("RC_BENCH_FORCE_DIST"))  # force: exercise the RCCL path on one GPU
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        with stdout_to_stderr():      # RCCL's version banner (C stdout, first communicator) must not land beside the JSON line
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()

    import redclust_amd as rc
    n, K = N_POINTS, N_CLUST
    data = rc.generatemixture(n, K, seed=1)          # same data on every rank; chains differ by seed
    D, truth = data["distancematrix"], data["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    r, p = 1.0, 0.5
    chain_seed = 1 + rank                             # chain seeds 1..N

    ctx = rc.Context(D, device=local_rank, kcap=max(128, 2 * K), storage_bits=BITS)
    ctx.set_params(**P)
    ctx.set_state(truth)                              # stationary regime: generating labels
    ctx.cocluster_reset()

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    sweep = 0
    # Device settle (untimed, before the W warm-up steps): a short window right after the idle set-up phase does not see the clocks
    # of a running chain — 20 timed sweeps measured 13.2 k sweeps/s after a host-side pause, 12.1 k over the following 200 sweeps and
    # 14.3 k once a second of sweeps had run (tools/ramp_probe.py) — so the same workload runs for RC_BENCH_SETTLE_MS (default 400,
    # 0 = off; reported as config.settle_ms) before the warm-up and the timed steps.
    settle_ms = float(os.environ.get("RC_BENCH_SETTLE_MS", 400))
    unsettled = None
    if settle_ms > 0:
        # the same W + K steps once BEFORE the settle phase (no event timing), reported beside the headline as
        # config.sweeps_per_s_before_settle: what the window measures straight after the idle set-up phase
        for _ in range(args.warmup):
            ctx.gibbs_sweep(r, p, chain_seed, sweep, blocking=False); sweep += 1
        ctx.synchronize()
        t_u = time.perf_counter()
        for _ in range(args.steps):
            ctx.gibbs_sweep(r, p, chain_seed, sweep, blocking=False); sweep += 1
        ctx.synchronize()
        unsettled = args.steps / (time.perf_counter() - t_u)
        t_s = time.perf_counter()
        while (time.perf_counter() - t_s) * 1e3 < settle_ms:
            for _ in range(256):
                ctx.gibbs_sweep(r, p, chain_seed, sweep, blocking=False)
                sweep += 1
            ctx.synchronize()
    for _ in range(args.warmup):
        ctx.gibbs_sweep(r, p, chain_seed, sweep, blocking=False)
        sweep += 1
    ctx.synchronize()
    # HIP events around the row-reduction kernel, on the stream it is launched on: every launch for short runs (the driver's
    # --steps 20), every 4th otherwise (an event pair costs the stream a few microseconds)
    time_every = int(os.environ.get('RC_BENCH_TIME_EVERY', 1 if args.steps < 64 else 4))
    ctx.kernel_timing(enable=0 if os.environ.get('RC_BENCH_NO_TIMING') else time_every)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.gibbs_sweep(r, p, chain_seed, sweep, blocking=False)
        sweep += 1
    ctx.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    bulk_ms, bulk_launches = ctx.kernel_timing(enable=0)
    stats = ctx.sweep_stats()

    # same workload in the exact incremental mode (no row reduction while labels are stable) — reported as an extra
    inc_sweeps_per_s = None
    if not os.environ.get("RC_BENCH_NO_INCREMENTAL"):
        ctx.set_mode("incremental")
        for _ in range(args.warmup):
            ctx.gibbs_sweep(r, p, chain_seed, sweep, blocking=False); sweep += 1
        ctx.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            ctx.gibbs_sweep(r, p, chain_seed, sweep, blocking=False); sweep += 1
        ctx.synchronize()
        inc_sweeps_per_s = args.steps / (time.perf_counter() - t1)
        ctx.set_mode("full")

    # recorded sample + the one collective of the path: sum all-reduce of the integer co-clustering counts, through the
    # LIBRARY's communicator (rc_comm_create / rc_comm_allreduce_counts: RCCL inside libredclust_hip.so; torch.distributed
    # only carries the 128-byte unique id).  Also at N = 1: a real communicator of size 1.  If the library path cannot be
    # set up on every rank the counts are merged with torch.distributed instead and the line says so.
    ctx.record_sample(False)
    allreduce_ms, merge_path = None, None
    lib_ok = 1
    try:
        rc.Comm.unique_id()                                  # opens librccl: every rank checks before anyone commits
    except Exception as e:                                   # noqa: BLE001
        lib_ok, merge_path = 0, f"torch.distributed all_reduce (library RCCL unavailable: {e})"
    if distributed:
        flag = torch.tensor([lib_ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        lib_ok = int(flag.item())
    total_samples = None
    if lib_ok:
        # RCCL prints a version banner on the C stdout when its first communicator comes up: keep this process's stdout
        # for the one JSON line (fd 1 points at stderr while the communicator is built)
        with stdout_to_stderr():
            total_samples, allreduce_ms = rc.library_merge(ctx, local_rank, 1)
        merge_path = "libredclust_hip.so rc_comm_allreduce_counts (RCCL ncclAllReduce sum uint32, in place)"
    elif distributed:
        counts = rc.device_counts_tensor(ctx, local_rank)   # zero-copy view of the library's device buffer
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        allreduce_ms = (time.perf_counter() - t1) * 1e3
        total_samples = world
        merge_path = merge_path or "torch.distributed all_reduce (library RCCL unavailable on some rank)"
    if distributed:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    # every chain recorded one sample: each point co-clusters with itself once per chain, and the matrix is symmetric
    counts = rc.device_counts_tensor(ctx, local_rank)
    diag_ok = (bool((counts.diagonal() == world).all().item()) and bool((counts[:, :n] == counts[:, :n].T).all().item())
               and (total_samples in (None, world)))

    # Second figure ("exercises movement"): the same N, K with overlapping clusters (sigma = 0.2 instead of 0.1:
    # about 0.5 % of the labels move per sweep and clusters are born and die), burn-in from the generating labels excluded from the timing.
    moving = None
    if rank == 0 and not os.environ.get("RC_BENCH_NO_MOVING"):
        sig = float(os.environ.get("RC_BENCH_MOVING_SIGMA", 0.2))
        dm = rc.generatemixture(n, K, seed=2, sigma=sig)
        Dm, tm = dm["distancematrix"], dm["clusts"]
        Pm = rc.likelihood_hyperparams(Dm, tm)
        cm = rc.Context(Dm, device=local_rank, kcap=max(512, 8 * K), storage_bits=BITS)
        cm.set_params(**Pm)
        cm.set_state(tm)
        sw = 0
        for _ in range(60):                               # burn-in to the moving equilibrium
            cm.gibbs_sweep(r, p, 7, sw, blocking=False); sw += 1
        cm.synchronize()
        msteps = max(20, min(args.steps, 200))
        ch = rounds = 0
        t1 = time.perf_counter()
        for _ in range(msteps):
            cm.gibbs_sweep(r, p, 7, sw, blocking=True); sw += 1   # blocking: the change count of every sweep is read
            st = cm.sweep_stats(); ch += st["n_changes"]; rounds += st["n_rounds"]
        t_block = time.perf_counter() - t1
        t1 = time.perf_counter()
        for _ in range(msteps):
            cm.gibbs_sweep(r, p, 7, sw, blocking=False); sw += 1
        cm.synchronize()
        t_async = time.perf_counter() - t1
        # SURVEY.md §8(d)'s other movement workload: labels uniform on 1..K from a fixed seed on the headline data — the first sweeps
        # relabel nearly every point (a one-off transient: thousands of changes resolved in batches)
        uni = np.random.default_rng(13).integers(1, K + 1, size=n).astype(np.int64)
        cu = rc.Context(D, device=local_rank, kcap=max(256, 4 * K), storage_bits=BITS)
        cu.set_params(**P); cu.set_state(uni); cu.synchronize()
        uniform_init = []
        for q in range(3):
            t1 = time.perf_counter(); cu.gibbs_sweep(r, p, 7, q, blocking=True); t_q = time.perf_counter() - t1
            st = cu.sweep_stats()
            uniform_init.append({"sweep": q, "ms": t_q * 1e3, "label_changes": st["n_changes"], "resolve_rounds": st["n_rounds"], "K": st["K"]})
        cu.close()
        moving = {"sigma": sig, "uniform_init_first_sweeps": uniform_init, "sweeps_per_s": msteps / t_async, "ms_per_sweep": t_async / msteps * 1e3,
                  "sweeps_per_s_blocking": msteps / t_block, "label_changes_per_sweep": ch / msteps,
                  "resolve_rounds_per_sweep": rounds / msteps, "K": cm.sweep_stats()["K"], "steps": msteps,
                  "note": "overlapping clusters, equilibrium after 60 burn-in sweeps from the generating labels"}
        cm.close()
        del Dm

    # Third figure: the whole iteration of the reference with its DEFAULT options (MCMCOptionsList(): numMH = 1, numGibbs = 5,
    # src/types.jl:3-43) — sample_r, sample_p, one split-merge proposal, the Gibbs sweep — through rc_run_chain
    defaults = None
    if rank == 0 and not os.environ.get("RC_BENCH_NO_DEFAULTS"):
        cd = rc.Context(D, device=local_rank, kcap=max(128, 2 * K), storage_bits=BITS)
        cd.set_params(**P); cd.set_state(truth); cd.cocluster_reset()
        cd.attach_host_matrices(D)                        # the proposals' restricted scans read the host matrix (logD derived by the library)
        cd.run_chain(100, 0, 10, 5, 1, 1, r, p, 1.0)               # warm-up (worker threads, pinned buffers, caches)
        its = 1000
        t1 = time.perf_counter()
        chd = cd.run_chain(its, 0, 10, 5, 1, 1, r, p, 1.0, first_iter=100)
        t_def = time.perf_counter() - t1
        defaults = {"numMH": 1, "numGibbs": 5, "iterations": its, "iterations_per_s": its / t_def, "ms_per_iteration": t_def / its * 1e3,
                    "splitmerge_acceptances": int(chd["splitmerge_acceptances"].sum()), "splitmerge_splits": int(chd["splitmerge_splits"].sum()),
                    "note": "rc_run_chain, speculative split-merge pipeline (proposals of several iterations decided concurrently on host "
                            "threads; bit-identical to the sequential loop)"}
        cd.close()

    if rank == 0:
        # bytes of matrix data the selected row-reduction kernel has to read per sweep: k_bulk reads every entry of D
        # and logD (2·n²·sizeof); k_bulk_sym exploits symmetry and reads the upper triangle only
        kernel_family, alg_bytes = ctx.bulk_kernel_info()     # bytes the kernel that ran has to read per launch
        kernel_name = ctx.bulk_kernel_name()
        esz = BITS / 8.0
        # logD derived on the fly (no logD given, 64-bit storage): one matrix is read instead of two
        derived = alg_bytes < 1.5 * (n * (n + 1) / 2 if kernel_family != "k_bulk" else n * n) * esz
        # a sweep is priced at 2·n²·sizeof (D and logD read once each), or n²·sizeof "if the build
        # recomputes log on the fly instead of staging logD"
        survey_bytes = (1.0 if derived else 2.0) * n * n * esz
        full_bytes = 2.0 * n * n * esz
        value = world * args.steps / dt
        bulk_avg_ms = bulk_ms / max(bulk_launches, 1)
        achieved = survey_bytes / (bulk_avg_ms * 1e-3) / 1e9 if bulk_launches else None       # §8(d) bytes ÷ kernel time
        achieved_read = alg_bytes / (bulk_avg_ms * 1e-3) / 1e9 if bulk_launches else None     # bytes actually read ÷ kernel time
        # HBM bytes per launch of that kernel from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, gfx950 FETCH_SIZE
        # correction applied): collected OFFLINE with this same command (tools/prof_r02.sh) and committed — not measured in
        # this run; `traffic_source` names the file
        traffic = traffic_source = None
        tag = kernel_name.replace("<", "_").replace(">", "").replace(", ", "_").replace(" ", "")
        for rnd in ("r02", "r01"):
            for name in (f"pmc_traffic_n{n}_{tag}.json", f"pmc_traffic_n{n}_{kernel_family}.json"):
                f = os.path.join(ROOT, "profiles", rnd, name)
                if traffic is None and os.path.exists(f) and BITS == 64 and (name.endswith(f"{tag}.json") or not derived):
                    traffic = json.load(open(f))["k_bulk_hbm_bytes_per_launch"]
                    traffic_source = f"profiles/{rnd}/{name} (rocprofv3 --pmc, collected offline with the same command)"
        ceiling = None
        try:   # the fraction is also reported against a streaming-read ceiling measured on this box, now
            ceiling = rc.measure_read_ceiling(local_rank, 2048, 5)
        except Exception:   # noqa: BLE001
            pass
        out = {
            "metric": "Gibbs sweeps/sec (n×n distM)", "value": value, "unit": "sweeps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": f"i{BITS} fixed-point storage, i64 exact sums, f64 scores",
            "data": "synthetic",
            "config": {"workload": f"generatemixture N={n} K={K} dim={K} sigma=0.1 dense Float64 distM ({BITS}-bit fixed-point storage), 1 chain per GPU, "
                                   "numMH=0 Gibbs sweep, init = generating labels (stationary), r=1 p=0.5",
                       "chains": world, "n": n, "K": K, "parallelism": f"chains x{world}",
                       "settle_ms": settle_ms, "sweeps_per_s_before_settle": unsettled, "settle_note": "untimed sweeps of the same workload before the warm-up steps (device clocks of a running chain; RC_BENCH_SETTLE_MS=0 disables)"},
            "logD": "derived on the fly (table log of the fixed-point D)" if derived else "stored",
            "sweep_GBps_algorithmic": value / world * survey_bytes / 1e9,          # whole sweep (not just the kernel) at §8(d) bytes
            "sweep_frac_of_hbm_peak": value / world * survey_bytes / 1e9 / HBM_PEAK_GBPS,
            "sweep_GBps_on_bytes_read": value / world * alg_bytes / 1e9,
            "sweep_GBps_vs_reference_dataflow": value / world * full_bytes / 1e9,  # 2·n²·sizeof per sweep, what the reference reads
            "label_changes_last_sweep": stats["n_changes"], "K_final": stats["K"],
            "moving_regime": moving,
            "reference_default_options": defaults,
            "incremental_mode_sweeps_per_s_rank0": inc_sweeps_per_s,
            "coclustering_allreduce_ms": allreduce_ms, "coclustering_merge_path": merge_path, "coclustering_diag_ok": diag_ok,
            # roofline of the dominant kernel.  `achieved` / `frac` use the algorithmic bytes SURVEY.md §8(d) prescribes
            # (see survey_bytes above).  The kernel itself reads less than that — only the upper triangle of the
            # symmetric matrix — so the physical figures (bytes it must read ÷ time) are given beside them, and
            # `traffic` is the HBM bytes per launch measured with the PMC counters.
            "roofline": {"kernel": kernel_name, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "measured_streaming_read_GBps": ceiling, "frac_of_measured_streaming_read": (achieved / ceiling) if achieved and ceiling else None,
                         "frac_on_bytes_read_of_measured_streaming_read": (achieved_read / ceiling) if achieved_read and ceiling else None,
                         "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic, "traffic_source": traffic_source,
                         "pricing": "achieved / frac: SURVEY.md §8(d) algorithmic bytes per sweep (n²·sizeof when logD is derived on the fly, "
                                    "2·n²·sizeof when stored) ÷ mean launch duration; *_on_bytes_read: the bytes this kernel has to read "
                                    "(the upper triangle only) ÷ the same duration",
                         "timed_every_nth_launch": time_every,
                         "avg_launch_ms": bulk_avg_ms, "launches": bulk_launches,
                         "algorithmic_bytes_per_launch": survey_bytes,
                         "bytes_read_by_kernel_per_launch": alg_bytes,
                         "achieved_on_bytes_read": achieved_read,
                         "frac_on_bytes_read": (achieved_read / HBM_PEAK_GBPS) if achieved_read else None,
                         "event_pair_overhead_ms_subtracted": ctx.event_overhead_ms()},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(D, P, truth, r, p)
        print(json.dumps(out))
    ctx.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
