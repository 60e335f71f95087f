#!/usr/bin/env python3
"""bench.py — Gibbs sweeps/sec of the MI355X sweep on BASELINE.json's roofline configuration.

  python bench.py --gpus N --steps K --warmup W          (N=1: plain python; N>1: one rank per GPU under
  python -m torch.distributed.run --nproc-per-node N …    torch.distributed, backend nccl = RCCL)

A "step" is one Gibbs sweep (sample_labels_Gibbs!, /root/reference/src/mcmc.jl:158-256) of one chain over the
synthetic N=8192, K=50 dense Float64 dissimilarity matrix (BASELINE.json configs[2], the configuration the
north_star's roofline target is quoted on); D and logD are already resident in HBM when the timed region
starts.  Chains are independent: with N GPUs every rank runs its own chain (weak scaling) and the only
collective is the final sum all-reduce of the n×n co-clustering counts (outside the timed region, reported).

Prints ONE JSON line on rank 0 (contract in the task statement) with
  `roofline`      the dominant kernel of the headline workload, k_bulk_syml2 (row reduction over the 48-bit packed upper triangle
                  of D, logD derived with a table log): PHYSICAL figures only — the bytes the kernel reads ÷ its mean launch duration
                  (HIP events carried in the dispatch, on the library's own stream) ÷ the 8 TB/s HBM peak; `bound` names what
                  limits it (VALU issue + memory-side atomics: its 201 MB working set is served from the 256 MiB Infinity Cache,
                  not from HBM), `compute` prices its VALU instruction stream, `equivalent_dataflow_GBps` keeps SURVEY.md §8(d)'s
                  algorithmic pricing (never as a fraction: the kernel reads 3/8 of those bytes);
  `other_configs` the other single-GPU BASELINE configurations, timed in this run: config 5 (N = 32768, K = 200, 32-bit storage, built
                  from points — HBM-resident: a real HBM roofline for k_bulk_sym32) and config 2 (N = 2000, K = 20: sweeps/s and
                  rc_run_chain iterations/s for numiters = 10000);
  `cpu_baseline`  the C restatement of the reference's loop, faithful-cost mode, one host core, bounded sample.
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args(argv)


def launch_plan(args, environ):
    """How this invocation runs (decided BEFORE torch is imported or any GPU call is made):
      "ranks"   --gpus N > 1 and no WORLD_SIZE: this process is only a launcher — it starts a FRESH child,
                `python -m torch.distributed.run --nproc-per-node N bench.py …` (one rank per GPU over RCCL), relays the child's
                stdout (rank 0's JSON line) and exits with its code.  Never an exec, never after HIP is initialised.
      "single"  RC_BENCH_SINGLE_PROCESS=1: N contexts on N devices driven by N host threads of this one process — the shape of
                rc_run_chains (the C-ABI path of the Julia glue).
      "rank"    one rank of a torch.distributed.run job (WORLD_SIZE set; must equal --gpus), or the plain 1-GPU run.
    Raises SystemExit(2) when WORLD_SIZE and --gpus disagree: a SCALE run must never silently measure fewer GPUs."""
    single = bool(environ.get("RC_BENCH_SINGLE_PROCESS")) and environ.get("RC_BENCH_SINGLE_PROCESS") != "0"
    world_env = environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if single:
        if world_env not in (None, "1"):
            raise SystemExit("bench.py: RC_BENCH_SINGLE_PROCESS=1 drives all GPUs from one process; do not launch it under torch.distributed.run")
        return "single"
    if world_env is None:
        return "ranks" if args.gpus > 1 else "rank"
    if int(world_env) != args.gpus:
        sys.stderr.write(f"bench.py: WORLD_SIZE={world_env} but --gpus {args.gpus}: refusing to report a figure for the wrong number of GPUs\n")
        raise SystemExit(2)
    return "rank"


def launch_ranks(args, argv):
    """--gpus N without a launcher: start N ranks in a fresh child process group and relay rank 0's line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    out = proc.stdout.decode(errors="replace")
    lines = [ln for ln in out.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if proc.returncode == 0 and len(lines) == 1:
        print(lines[0])
        return 0
    sys.stderr.write(out)
    sys.stderr.write(f"bench.py: the {args.gpus}-rank child exited with code {proc.returncode} and printed {len(lines)} result line(s)\n")
    return proc.returncode or 1


def timed_windows(ctx, seed, r, p, steps, warmup, windows, sync_all, sweep0=0):
    """W warm-up sweeps, then `windows` windows of EXACTLY `steps` sweeps each, every window bracketed by sync_all() (barrier over
    all chains + device synchronise) on both sides.  Returns (per-window seconds, next sweep index)."""
    sweep = sweep0
    for _ in range(warmup):
        ctx.gibbs_sweep(r, p, seed, sweep, blocking=False); sweep += 1
    ctx.synchronize()
    times = []
    for _ in range(windows):
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.gibbs_sweep(r, p, seed, sweep, blocking=False); sweep += 1
        ctx.synchronize()
        sync_all()
        times.append(time.perf_counter() - t0)
    return times, sweep


def settle(ctx, seed, r, p, ms, sweep):
    t_s = time.perf_counter()
    while (time.perf_counter() - t_s) * 1e3 < ms:
        for _ in range(256):
            ctx.gibbs_sweep(r, p, seed, sweep, blocking=False); sweep += 1
        ctx.synchronize()
    return sweep


def median(xs):
    xs = sorted(xs)
    m = len(xs) // 2
    return xs[m] if len(xs) % 2 else 0.5 * (xs[m - 1] + xs[m])


def profile_json(*names):
    """The newest committed profiles/rNN/<name> among `names` (offline rocprofv3 --pmc results this run cannot collect itself)."""
    for rnd in ("r04", "r03", "r02", "r01"):
        for name in names:
            f = os.path.join(ROOT, "profiles", rnd, name)
            if os.path.exists(f):
                try:
                    return json.load(open(f)), f"profiles/{rnd}/{name}"
                except Exception:   # noqa: BLE001
                    pass
    return None, None


def kernel_roofline(ctx, rc, steps, sweep_fn, traffic_names, counters_names, kernel_timing_every=1):
    """Physical roofline of the row-reduction kernel the context runs: bytes it has to read (rc_bulk_kernel_info) ÷ its mean launch
    duration (HIP events in the dispatch) ÷ 8 TB/s.  sweep_fn(k) enqueues k sweeps and synchronises."""
    sweep_fn(max(8, steps // 4))                                   # warm-up
    ctx.kernel_timing(enable=kernel_timing_every)
    t0 = time.perf_counter()
    sweep_fn(steps)
    dt = time.perf_counter() - t0
    bulk_ms, launches = ctx.kernel_timing(enable=0)
    fam, nbytes = ctx.bulk_kernel_info()
    avg_ms = bulk_ms / max(launches, 1)
    achieved = nbytes / (avg_ms * 1e-3) / 1e9 if launches else None
    traffic, src = profile_json(*traffic_names)
    return {"kernel": ctx.bulk_kernel_name(), "bytes_read_per_launch": nbytes, "avg_launch_ms": avg_ms, "launches": launches,
            "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBPS) if achieved else None,
            "per_sweep_period": {"ms": dt / steps * 1e3, "GBps": nbytes / (dt / steps) / 1e9, "frac": nbytes / (dt / steps) / 1e9 / HBM_PEAK_GBPS},
            "traffic": traffic["k_bulk_hbm_bytes_per_launch"] if traffic else None, "traffic_source": src}, dt


def config5_leg(rc, dev0, kcap, steps):
    """BASELINE config 5 on one GPU: N = 32768, K = 200, 32-bit fixed-point storage, built from the points on the device (no n x n host
    matrix).  2 x 2.15 GB (upper triangles of D and logD) are read per sweep: far beyond the 256 MiB Infinity Cache, so this is the
    workload on which an HBM roofline is meaningful."""
    n, K = 32768, 200
    t0 = time.perf_counter()
    data = rc.generatemixture(n, K, seed=1, points_only=True)
    pts, truth = data["points"], data["clusts"]
    ctx = rc.Context.from_points(pts, kcap=kcap, storage_bits=32, device=dev0)
    P = rc.likelihood_hyperparams_device(ctx, truth)
    ctx.set_params(**P); ctx.set_state(truth)
    setup_s = time.perf_counter() - t0
    sweep = [0]

    def run(k):
        for _ in range(k):
            ctx.gibbs_sweep(1.0, 0.5, 1, sweep[0], blocking=False); sweep[0] += 1
        ctx.synchronize()
    roof, _ = kernel_roofline(ctx, rc, steps, run, ("pmc_traffic_n32768_k_bulk_sym32.json",), ())
    tw = []
    for _ in range(3):
        t1 = time.perf_counter(); run(steps); tw.append(time.perf_counter() - t1)
    dt = median(tw)
    st = ctx.sweep_stats()
    out = {"workload": f"generatemixture N={n} K={K} dim={K} sigma=0.1 from points (rc_create_from_points), 32-bit fixed-point storage of D and logD, "
                       "numMH=0 Gibbs sweep, init = generating labels, r=1 p=0.5, library-default capacity",
           "sweeps_per_s": steps / dt, "ms_per_sweep": dt / steps * 1e3, "steps": steps, "windows_sweeps_per_s": [steps / x for x in tw],
           "label_changes_last_sweep": st["n_changes"], "K_final": st["K"], "capacity": ctx.capacity_info(), "setup_s": setup_s,
           "dtype": "i32 fixed-point storage of D and logD (2^-30 of the largest entry), i64 exact sums, f64 scores",
           "roofline": dict(roof, bound="hbm", note="k_bulk_sym32 reads the upper triangles of D and logD (n(n+1)/2 x 4 B each = 4.29 GB per sweep): HBM-resident.  "
                                                       "Its launches alternate between two streams and each runs three blocks per CU beside the resolver: a launch's "
                                                       "event-timed duration includes the time its blocks wait for those of the previous launch to retire, so `frac` "
                                                       "(per launch) understates and `per_sweep_period.frac` is the sustained rate")}
    ctx.close()
    return out


def config2_leg(rc, dev0, kcap, steps):
    """BASELINE config 2: N = 2000, K = 20 dense Float64 distM, one chain, numiters = 10000 — the sweep alone (as `value`) and the whole
    iteration of runsampler through rc_run_chain (sample_r, sample_p, [split-merge,] sweep, recording with the reference's default
    burn-in and thinning), in the library-default full mode and in the incremental mode the hosts use."""
    n, K, numiters = 2000, 20, 10000
    data = rc.generatemixture(n, K, seed=1)
    D, truth = data["distancematrix"], data["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    ctx = rc.Context(D, device=dev0, kcap=kcap)
    ctx.set_params(**P); ctx.set_state(truth); ctx.cocluster_reset()
    tw, _ = timed_windows(ctx, 1, 1.0, 0.5, max(steps, 200), 50, 3, lambda: None)
    out = {"workload": f"generatemixture N={n} K={K} dim={K} sigma=0.1 dense Float64 distM, 1 chain, init = generating labels",
           "sweeps_per_s": max(steps, 200) / median(tw), "kernel": ctx.bulk_kernel_name(), "numiters": numiters}
    ctx.attach_host_matrices(D)
    for name, numMH, mode in (("numMH0", 0, "full"), ("numMH0_incremental_mode", 0, "incremental"), ("reference_defaults_numMH1", 1, "full"),
                              ("reference_defaults_numMH1_incremental_mode", 1, "incremental")):
        ctx.set_state(truth); ctx.cocluster_reset(); ctx.set_mode(mode)
        ctx.run_chain(200, 0, 1, 5, numMH, 1, 1.0, 0.5, 1.0)                      # warm-up (worker threads, pinned buffers)
        t1 = time.perf_counter()
        ch = ctx.run_chain(numiters, numiters // 5, 1, 5, numMH, 1, 1.0, 0.5, 1.0, first_iter=200)   # MCMCOptionsList defaults: burnin = numiters / 5, thin = 1
        dt = time.perf_counter() - t1
        cs = ctx.chain_stats()
        out[name] = {"iterations_per_s": numiters / dt, "seconds_for_numiters": dt, "samples": int(ch["num_samples"]), "numMH": numMH, "numGibbs": 5, "mode": mode,
                     "K_final": int(ch["K"][-1]), "workers": cs["workers"], "rollbacks": cs["rollbacks"],
                     "splitmerge_acceptances": int(ch["splitmerge_acceptances"].sum()) if numMH else 0}
    ctx.set_mode("full")
    ctx.close()
    return out


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    plan = launch_plan(args, os.environ)
    if plan == "ranks":
        sys.exit(launch_ranks(args, argv))

    import threading
    import torch
    single = plan == "single"
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = args.gpus                                                   # == WORLD_SIZE (launch_plan), or the threads of the single-process mode
    distributed = (not single) and (world > 1 or bool(os.environ.get("RC_BENCH_FORCE_DIST")))  # force: exercise the RCCL path on one GPU
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    need = world if single else local_rank + 1
    if ndev < need or (not single and world > ndev):
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) are visible here: refusing to report a figure for fewer GPUs than asked for\n")
        raise SystemExit(3)
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        with stdout_to_stderr():      # RCCL's version banner (C stdout, first communicator) must not land beside the JSON line
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        assert dist.get_world_size() == args.gpus

    import redclust_amd as rc
    n, K = N_POINTS, N_CLUST
    data = rc.generatemixture(n, K, seed=1)          # same data on every rank; chains differ by seed
    D, truth = data["distancematrix"], data["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    r, p = 1.0, 0.5
    kcap = int(os.environ.get("RC_BENCH_KCAP", 0))    # 0 = the library's default (automatic capacity): what runsampler and the Julia glue use
    windows = max(1, int(os.environ.get("RC_BENCH_WINDOWS", 5)))
    settle_ms = float(os.environ.get("RC_BENCH_SETTLE_MS", 400))
    time_every = int(os.environ.get('RC_BENCH_TIME_EVERY', 1 if args.steps < 64 else 4))

    # the chains of this process: one (its rank's) under torch.distributed.run, `world` of them in the single-process mode
    devices = list(range(world)) if single else [local_rank]
    seeds = [1 + d for d in devices] if single else [1 + rank]          # chain seeds 1..N

    def make_ctx(dev):
        c = rc.Context(D, device=dev, kcap=kcap, storage_bits=BITS)
        c.set_params(**P)
        c.set_state(truth)                            # stationary regime: generating labels
        c.cocluster_reset()
        return c
    ctxs = [make_ctx(d) for d in devices]
    ctx = ctxs[0]

    tbar = threading.Barrier(len(ctxs)) if single and len(ctxs) > 1 else None

    def sync_all():
        if tbar is not None:
            tbar.wait()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    # Per chain: the W + K steps once BEFORE the settle phase (reported as config.sweeps_per_s_before_settle), a settle phase
    # (untimed: a short window right after the idle set-up does not see the clocks of a running chain — tools/ramp_probe.py;
    # RC_BENCH_SETTLE_MS=0 turns both off), then W warm-up steps and RC_BENCH_WINDOWS (5) timed windows of exactly K steps.
    # HIP events ride in the dispatch of the row-reduction launches of the timed windows (every launch for --steps < 64).
    results = [None] * len(ctxs)

    def chain_job(ci):
        try:
            chain_job_body(ci)
        except BaseException:      # noqa: BLE001 — a failed chain must not leave the others waiting at the window barrier for ever
            import traceback
            traceback.print_exc()
            if tbar is not None:
                tbar.abort()

    def chain_job_body(ci):
        c, sd = ctxs[ci], seeds[ci]
        sweep, unsettled = 0, None
        if settle_ms > 0:
            tu, sweep = timed_windows(c, sd, r, p, args.steps, args.warmup, 1, sync_all, sweep)
            unsettled = tu[0]
            sweep = settle(c, sd, r, p, settle_ms, sweep)
        for _ in range(args.warmup):
            c.gibbs_sweep(r, p, sd, sweep, blocking=False); sweep += 1
        c.synchronize()
        c.kernel_timing(enable=0 if os.environ.get('RC_BENCH_NO_TIMING') else time_every)
        tw, sweep = timed_windows(c, sd, r, p, args.steps, 0, windows, sync_all, sweep)
        bulk_ms, bulk_launches = c.kernel_timing(enable=0)
        results[ci] = dict(times=tw, unsettled=unsettled, bulk_ms=bulk_ms, bulk_launches=bulk_launches, sweep=sweep, stats=c.sweep_stats())

    if len(ctxs) == 1:
        chain_job(0)
    else:
        th = [threading.Thread(target=chain_job, args=(ci,)) for ci in range(len(ctxs))]
        for t in th: t.start()
        for t in th: t.join()
        if any(x is None for x in results):
            raise SystemExit("bench.py: a chain thread failed")
    # a window's time is the MAX over the chains (threads here, ranks below)
    win = [max(res["times"][w] for res in results) for w in range(windows)]
    unsettled_t = max(res["unsettled"] for res in results) if settle_ms > 0 else None
    if distributed:
        tmax = torch.tensor(win + [unsettled_t or 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        vals = [float(x) for x in tmax.tolist()]
        win, unsettled_t = vals[:windows], (vals[windows] if settle_ms > 0 else None)
    dt = median(win)
    bulk_ms, bulk_launches = results[0]["bulk_ms"], results[0]["bulk_launches"]
    stats, sweep = results[0]["stats"], results[0]["sweep"]
    chain_seed = seeds[0]

    # same workload in the exact incremental mode (no row reduction while labels are stable) — reported as an extra
    inc_sweeps_per_s = None
    if not os.environ.get("RC_BENCH_NO_INCREMENTAL"):
        ctx.set_mode("incremental")
        ti, sweep = timed_windows(ctx, chain_seed, r, p, args.steps, args.warmup, 1, lambda: None, sweep)
        inc_sweeps_per_s = args.steps / ti[0]
        ctx.set_mode("full")

    # recorded sample + the one collective of the path: sum all-reduce of the integer co-clustering counts, through the
    # LIBRARY's communicator (rc_comm_create / rc_comm_allreduce_counts: RCCL inside libredclust_hip.so; torch.distributed
    # only carries the 128-byte unique id; all ranks agree on the path first — chains.agreed_merge).  Also at N = 1: a real
    # communicator of size 1.  Single-process mode: one communicator over all devices (ncclCommInitAll), as rc_run_chains.
    for c in ctxs:
        c.record_sample(False)
    with stdout_to_stderr():   # RCCL prints a version banner on the C stdout when its first communicator comes up
        if single:
            comm = rc.Comm(devices)
            try:
                total_samples, allreduce_ms = comm.allreduce_counts(ctxs, [1] * len(ctxs))
            finally:
                comm.close()
            merge_path = "libredclust_hip.so rc_comm_allreduce_counts over ncclCommInitAll (one process, one context per device)"
            nranks = len(devices)
        else:
            total_samples, allreduce_ms, merge_path = rc.agreed_merge(ctx, local_rank, 1)
            nranks = dist.get_world_size() if distributed else 1
    # every chain recorded one sample: each point co-clusters with itself once per chain, and the matrix is symmetric
    counts = rc.device_counts_tensor(ctx, devices[0])
    diag_ok = (bool((counts.diagonal() == world).all().item()) and bool((counts[:, :n] == counts[:, :n].T).all().item())
               and total_samples == world)

    # ---- the other figures: rank 0 / chain 0 only, on its own device ---------------------------------------------------
    extras = rank == 0
    dev0 = devices[0]
    for c in ctxs[1:]:
        c.close()

    # The capacity the library's default gives vs. an explicit one: the context above IS the default (kcap = 0) unless
    # RC_BENCH_KCAP says otherwise; one window with 512 slots (round 2's default) beside it
    kcap_info = ctx.capacity_info()
    kcap512 = None
    if extras and not os.environ.get("RC_BENCH_NO_KCAP512"):
        c5 = rc.Context(D, device=dev0, kcap=512, storage_bits=BITS)
        c5.set_params(**P); c5.set_state(truth)
        t5, _ = timed_windows(c5, 1, r, p, args.steps, args.warmup + 200, 3, lambda: None)
        kcap512 = args.steps / median(t5)
        c5.close()

    # Second figure ("exercises movement"): the same N, K with overlapping clusters (sigma = 0.2 instead of 0.1:
    # about 0.5 % of the labels move per sweep and clusters are born and die), burn-in from the generating labels excluded from the timing.
    moving = None
    Dm = tm = Pm = None
    if extras and not os.environ.get("RC_BENCH_NO_MOVING"):
        sig = float(os.environ.get("RC_BENCH_MOVING_SIGMA", 0.2))
        dm = rc.generatemixture(n, K, seed=2, sigma=sig)
        Dm, tm = dm["distancematrix"], dm["clusts"]
        Pm = rc.likelihood_hyperparams(Dm, tm)
        cm = rc.Context(Dm, device=dev0, kcap=kcap, storage_bits=BITS)    # default capacity: grows with the chain
        cm.set_params(**Pm)
        cm.set_state(tm)
        sw = 0
        for _ in range(60):                               # burn-in to the moving equilibrium
            cm.gibbs_sweep(r, p, 7, sw, blocking=False); sw += 1
        cm.synchronize()
        msteps = max(100, min(args.steps, 200))           # (an extra leg, not `value`: 100 sweeps at least — 20 sweeps of this chain are 3 ms and 600-1000 label changes, too few for a stable rate)
        ch = rounds = 0
        t1 = time.perf_counter()
        for _ in range(msteps):
            cm.gibbs_sweep(r, p, 7, sw, blocking=True); sw += 1   # blocking: the change count of every sweep is read
            st = cm.sweep_stats(); ch += st["n_changes"]; rounds += st["n_rounds"]
        t_block = time.perf_counter() - t1
        tms, sw = timed_windows(cm, 7, r, p, msteps, 0, 3, lambda: None, sw)
        t_async = median(tms)
        # the same chain in the exact incremental mode (what runsampler and the Julia glue use: no row reduction beside the resolver)
        cm.set_mode("incremental")
        tmi, sw = timed_windows(cm, 7, r, p, msteps, 10, 3, lambda: None, sw)
        cm.set_mode("full")
        # SURVEY.md §8(d)'s other movement workload: labels uniform on 1..K from a fixed seed on the headline data — the first sweeps
        # relabel nearly every point (a one-off transient: thousands of changes resolved in batches)
        uni = np.random.default_rng(13).integers(1, K + 1, size=n).astype(np.int64)
        cu = rc.Context(D, device=dev0, kcap=kcap, storage_bits=BITS)
        cu.set_params(**P); cu.set_state(uni); cu.synchronize()
        uniform_init = []
        for q in range(3):
            t1 = time.perf_counter(); cu.gibbs_sweep(r, p, 7, q, blocking=True); t_q = time.perf_counter() - t1
            st = cu.sweep_stats()
            uniform_init.append({"sweep": q, "ms": t_q * 1e3, "label_changes": st["n_changes"], "resolve_rounds": st["n_rounds"], "K": st["K"]})
        uni_cap = cu.capacity_info()
        cu.close()
        moving = {"sigma": sig, "uniform_init_first_sweeps": uniform_init, "uniform_init_capacity": uni_cap,
                  "sweeps_per_s": msteps / t_async, "ms_per_sweep": t_async / msteps * 1e3,
                  "sweeps_per_s_windows": [msteps / x for x in tms],
                  "sweeps_per_s_incremental_mode": msteps / median(tmi),
                  "sweeps_per_s_blocking": msteps / t_block, "label_changes_per_sweep": ch / msteps,
                  "resolve_rounds_per_sweep": rounds / msteps, "K": cm.sweep_stats()["K"], "steps": msteps, "capacity": cm.capacity_info(),
                  "note": "overlapping clusters, equilibrium after 60 burn-in sweeps from the generating labels; library-default capacity"}
        cm.close()

    # Third figure: the whole iteration of the reference with its DEFAULT options (MCMCOptionsList(): numMH = 1, numGibbs = 5,
    # src/types.jl:3-43) — sample_r, sample_p, one split-merge proposal, the Gibbs sweep — through rc_run_chain: on the headline
    # data (separated clusters: proposals are rejected) and on the moving data (proposals get accepted: the speculative pipeline
    # rolls back), with the pipeline's rollbacks and worker threads
    defaults = None
    if extras and not os.environ.get("RC_BENCH_NO_DEFAULTS"):
        def default_options_leg(Dx, Px, labels, burn, mode="full", workers=0, its=1000):
            cd = rc.Context(Dx, device=dev0, kcap=kcap, storage_bits=BITS)
            cd.set_params(**Px); cd.set_state(labels); cd.cocluster_reset()
            cd.set_mode(mode)                             # "full": the library default (as `value`); "incremental": what runsampler and the Julia glue set
            cd.set_option("chain_workers", workers)       # 0 = automatic: the host's cores (minus one) up to 24
            cd.attach_host_matrices(Dx)                   # the proposals' restricted scans read the host matrix (logD derived by the library)
            cd.run_chain(burn, 0, 10, 5, 1, 1, r, p, 1.0)             # warm-up (worker threads, pinned buffers, caches; burn-in on the moving data)
            t1 = time.perf_counter()
            chd = cd.run_chain(its, 0, 10, 5, 1, 1, r, p, 1.0, first_iter=burn)
            t_def = time.perf_counter() - t1
            cs = cd.chain_stats()
            out = {"numMH": 1, "numGibbs": 5, "mode": mode, "iterations": its, "iterations_per_s": its / t_def, "ms_per_iteration": t_def / its * 1e3,
                   "splitmerge_acceptances": int(chd["splitmerge_acceptances"].sum()), "splitmerge_splits": int(chd["splitmerge_splits"].sum()),
                   "rollbacks": cs["rollbacks"], "splits_evaluated_offline": cs["split_evals"], "workers": cs["workers"],
                   "K_final": int(chd["K"][-1]) if len(chd["K"]) else None}
            cd.close()
            return out
        defaults = default_options_leg(D, P, truth, 100)
        defaults["note"] = ("rc_run_chain, speculative split-merge pipeline (proposals of several iterations decided concurrently; "
                            "bit-identical to the sequential loop); headline data, stationary")
        # the same leg with 2, 4, 8 and 24 worker threads: on an 8-GPU node every chain has cores / 8 of the host
        if not os.environ.get("RC_BENCH_NO_WORKER_SWEEP"):
            defaults["by_worker_threads"] = {}
            for w in (2, 4, 8, 24):
                leg = default_options_leg(D, P, truth, 100, workers=w, its=600)
                defaults["by_worker_threads"][str(w)] = {"iterations_per_s": leg["iterations_per_s"], "workers": leg["workers"]}
            defaults["host_cores"] = os.cpu_count()
        if Dm is not None:
            defaults["moving_data"] = default_options_leg(Dm, Pm, tm, 200)
            defaults["moving_data_incremental_mode"] = default_options_leg(Dm, Pm, tm, 200, "incremental")
        # ... and where proposals ARE accepted, so that the pipeline rolls back (an accepted proposal voids the iterations launched
        # behind it).  On the synthetic sets the Gibbs sweep repairs any labelling by itself and no proposal is ever accepted
        # (tried: clusters merged in pairs, cut in halves, sigma up to 0.5: tools/acc_probe.py), so this leg runs the reference's own
        # example data — paper dataset 1 (n = 100, tests/golden/paper_datasets.npz, extracted from data/example_datasets.h5) from
        # random labels, accepted proposals kept (splitmerge = "intended") — speculative against synchronous loop, same chain.
        try:
            z = np.load(os.path.join(ROOT, "tests", "golden", "paper_datasets.npz"))
            D1, lab1 = np.ascontiguousarray(z["D1"]), z["labels1"]
            P1 = rc.likelihood_hyperparams(D1, lab1)
            init1 = np.random.default_rng(1).integers(1, 11, size=100).astype(np.int64)
            legs = {}
            for name, env in (("speculative", None), ("synchronous", "0")):
                if env is None: os.environ.pop("RC_CHAIN_PIPELINE", None)
                else: os.environ["RC_CHAIN_PIPELINE"] = env
                cr = rc.Context(D1, device=dev0)
                cr.set_params(**P1); cr.set_state(init1); cr.cocluster_reset(); cr.attach_host_matrices(D1)
                its = 3000
                t1 = time.perf_counter()
                chr_ = cr.run_chain(its, 0, 10, 5, 1, 3, r, p, 1.0, splitmerge="intended")
                t_rb = time.perf_counter() - t1
                cs = cr.chain_stats()
                legs[name] = {"iterations_per_s": its / t_rb, "splitmerge_acceptances": int(chr_["splitmerge_acceptances"].sum()),
                              "splitmerge_splits": int(chr_["splitmerge_splits"].sum()), "rollbacks": cs["rollbacks"], "workers": cs["workers"], "K_final": int(chr_["K"][-1])}
                cr.close()
            os.environ.pop("RC_CHAIN_PIPELINE", None)
            # ... and where MANY are accepted: the first 500 iterations of the same data from ONE cluster and from random labels with
            # three proposals per iteration (splitmerge = "intended") — rollbacks, and what a rolled-back iteration costs (it is redone
            # by the synchronous path and the speculation restarts behind it) against the synchronous loop on the same chain
            hot = {}
            # "every_proposal_accepted": the repulsion-free model from ONE cluster with the reference's split-merge AS WRITTEN — every
            # proposal is a split, it is accepted, and quirk Q1 drops it again together with the iteration's sweep, so the
            # chain stays in one cluster and EVERY iteration rolls the pipeline back: the cost of the rollback path at a 100 % rate
            for start, init_h, Ph, smode in (("one_cluster", np.ones(100, np.int64), P1, "intended"), ("random_labels", init1, P1, "intended"),
                                             ("every_proposal_accepted", np.ones(100, np.int64), dict(P1, repulsion=False), "as_written")):
                legs_h = {}
                for name, pipe in (("speculative", 1), ("synchronous", 0)):
                    cr = rc.Context(D1, device=dev0)
                    cr.set_params(**Ph); cr.set_state(init_h); cr.cocluster_reset(); cr.attach_host_matrices(D1)
                    cr.set_option("chain_pipeline", pipe)
                    its = 500
                    t1 = time.perf_counter()
                    chh = cr.run_chain(its, 0, 10, 5, 3, 5, r, p, 1.0, splitmerge=smode)
                    t_h = time.perf_counter() - t1
                    cs = cr.chain_stats()
                    legs_h[name] = {"iterations_per_s": its / t_h, "seconds": t_h, "proposals": 3 * its, "splitmerge_acceptances": int(chh["splitmerge_acceptances"].sum()),
                                    "splitmerge_splits": int(chh["splitmerge_splits"].sum()), "rollbacks": cs["rollbacks"], "workers": cs["workers"], "K_final": int(chh["K"][-1])}
                    cr.close()
                a_, b_ = legs_h["speculative"], legs_h["synchronous"]
                rb = max(a_["rollbacks"], 1)
                hot[start] = dict(legs_h, acceptance_rate=a_["splitmerge_acceptances"] / (3 * 500),
                                  same_chain=a_["splitmerge_acceptances"] == b_["splitmerge_acceptances"] and a_["K_final"] == b_["K_final"],
                                  # time the pipelined loop spends beyond what the synchronous loop needs for the SAME iterations, per rollback (negative: still ahead)
                                  splitmerge=smode, repulsion=bool(Ph["repulsion"]),
                                  extra_ms_per_rollback_vs_synchronous=(a_["seconds"] - b_["seconds"]) / rb * 1e3)
            defaults["with_many_accepted_proposals"] = dict(hot, iterations=500, numMH=3, data="paper dataset 1 (n = 100)")
            defaults["with_accepted_proposals"] = dict(legs, iterations=3000, data="paper dataset 1 (n = 100), random initial labels, splitmerge='intended'",
                                                       same_chain=legs["speculative"]["splitmerge_acceptances"] == legs["synchronous"]["splitmerge_acceptances"]
                                                       and legs["speculative"]["K_final"] == legs["synchronous"]["K_final"])
        except Exception as e:   # noqa: BLE001
            defaults["with_accepted_proposals"] = {"error": str(e)}
    del Dm
    other = None
    if extras and not os.environ.get("RC_BENCH_NO_OTHER_CONFIGS"):
        other = {}
        for name, fn in (("config2_N2000_K20", lambda: config2_leg(rc, dev0, kcap, args.steps)),
                         ("config5_N32768_K200_32bit", lambda: config5_leg(rc, dev0, kcap, max(10, min(args.steps, 40))))):
            try:
                other[name] = fn()
            except Exception as e:   # noqa: BLE001 — reported in the line, never silently dropped
                other[name] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        # bytes of matrix data the selected row-reduction kernel has to read per launch: k_bulk_syml2 streams the 48-bit packed upper
        # triangle of D (n(n+1)/2 x 6 B) and derives logD on the fly; k_bulk_sym reads the upper triangles of both stored matrices;
        # k_bulk every entry of the matrices it reads
        kernel_family, read_bytes = ctx.bulk_kernel_info()
        kernel_name = ctx.bulk_kernel_name()
        esz = BITS / 8.0
        derived = read_bytes < 1.5 * (n * (n + 1) / 2 if kernel_family != "k_bulk" else n * n) * esz
        packed48 = kernel_name.startswith("k_bulk_syml2<true, true")
        # a sweep is priced at 2·n²·sizeof (D and logD read once each), or n²·sizeof "if the build recomputes log on
        # the fly instead of staging logD".  This design reads less than either (symmetry, 48-bit packing), so that pricing is an
        # EQUIVALENT-DATAFLOW rate, reported under that name and never as a fraction of a bandwidth peak.
        survey_bytes = (1.0 if derived else 2.0) * n * n * esz
        full_bytes = 2.0 * n * n * esz
        value = world * args.steps / dt
        period_s = dt / args.steps
        bulk_avg_ms = bulk_ms / max(bulk_launches, 1)
        achieved = read_bytes / (bulk_avg_ms * 1e-3) / 1e9 if bulk_launches else None        # bytes the kernel reads ÷ its mean launch duration
        tag = kernel_name.replace("<", "_").replace(">", "").replace(", ", "_").replace(" ", "")
        # rocprofv3 --pmc results of this same command, collected OFFLINE (tools/prof_r04.sh) and committed: bytes at the L2's memory
        # side per launch (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate passes) and the instruction counters of the kernel
        traffic_rec, traffic_source = profile_json(f"pmc_traffic_n{n}_{tag}.json", f"pmc_traffic_n{n}_{kernel_family}.json") if BITS == 64 else (None, None)
        counters, counters_source = profile_json(f"counters_n{n}_{tag}.json") if BITS == 64 else (None, None)
        working_set_fits_mall = read_bytes <= 256 * 2 ** 20
        if kernel_name.startswith("k_bulk_syml2") and working_set_fits_mall:
            bound = "valu_issue+memory_side_atomics"
            served = ("Infinity Cache: the %.0f MB the kernel reads per launch stay resident in the 256 MiB last-level cache from one sweep to the next "
                      "(FETCH_SIZE counts those hits at the L2's memory side); HBM itself is close to idle" % (read_bytes / 1e6))
        else:
            bound = "hbm"
            served = "HBM" if not working_set_fits_mall else "Infinity Cache / HBM"
        compute = None
        if counters and bulk_launches:
            # VALU issue time of one launch: wave instructions x measured cycles per instruction and SIMD ÷ (SIMDs x clock)
            # (tools/valu_rate.hip: 1.9 cycles for 32-bit ALU ops with >= 3 waves per SIMD, 3.0-3.9 for 64-bit integer, FP64, DPP
            # and conversion instructions; the kernel's mix is priced at the figure the counters file states)
            simds = 4 * 256
            issue_s = counters["SQ_INSTS_VALU"] * counters["cycles_per_valu_inst"] / (simds * counters["clock_ghz"] * 1e9)
            compute = {"valu_wave_insts_per_launch": counters["SQ_INSTS_VALU"], "salu_wave_insts_per_launch": counters.get("SQ_INSTS_SALU"),
                       "lds_bank_conflict_rate": (counters["SQ_LDS_BANK_CONFLICT"] / counters["SQ_LDS_IDX_ACTIVE"]) if counters.get("SQ_LDS_IDX_ACTIVE") else None,
                       "atomic_bytes_per_launch": counters.get("WRITE_SIZE_bytes"), "cycles_per_valu_inst": counters["cycles_per_valu_inst"],
                       "simds": simds, "clock_ghz": counters["clock_ghz"], "valu_issue_us_per_launch": issue_s * 1e6,
                       "frac_of_launch": issue_s / (bulk_avg_ms * 1e-3), "source": counters_source}
        ceiling = None
        try:   # also against a streaming-read ceiling measured on this box, now (2 GiB: HBM, not the cache)
            ceiling = rc.measure_read_ceiling(dev0, 2048, 5)
        except Exception:   # noqa: BLE001
            pass
        dtype = (f"u48-packed fixed-point D (quantum 2^-47 of the largest entry; the {BITS}-bit master copy stays in HBM), logD = degree-4 table log of it "
                 "rounded to fixed point (|error| <= 2^-eL + 2.6e-13 + entry rounding), i64 exact sums, f64 scores") if packed48 else \
                f"i{BITS} fixed-point storage of D" + (" (logD derived: table log)" if derived else " and logD") + ", i64 exact sums, f64 scores"
        out = {
            "metric": "Gibbs sweeps/sec (n×n distM)", "value": value, "unit": "sweeps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": f"generatemixture N={n} K={K} dim={K} sigma=0.1 dense Float64 distM ({BITS}-bit fixed-point storage), 1 chain per GPU, "
                                   "numMH=0 Gibbs sweep, init = generating labels (stationary), r=1 p=0.5",
                       "chains": world, "n": n, "K": K, "parallelism": f"chains x{world}",
                       "launch": {"rank": "one process per GPU (torch.distributed.run)", "single": "one process, one host thread + context per GPU (RC_BENCH_SINGLE_PROCESS=1)"}["single" if single else "rank"] if world > 1 else "one process, one GPU",
                       "rccl_ranks": nranks,
                       "slot_capacity": kcap_info, "slot_capacity_note": "kcap = 0 at rc_create (the library default, what runsampler and the Julia glue pass): sized from the first state, grows on demand" if kcap == 0 else f"RC_BENCH_KCAP={kcap}",
                       "value_is": f"median of {windows} windows of exactly {args.steps} steps (each bracketed by barrier + device synchronize)",
                       "windows_sweeps_per_s": [world * args.steps / x for x in win],
                       "windows_min_max_sweeps_per_s": [world * args.steps / max(win), world * args.steps / min(win)],
                       "timed_region_s": sum(win),
                       "settle_ms": settle_ms, "sweeps_per_s_before_settle": (world * args.steps / unsettled_t) if unsettled_t else None,
                       "settle_note": "untimed sweeps of the same workload before the warm-up steps (device clocks of a running chain; RC_BENCH_SETTLE_MS=0 disables)"},
            "logD": "derived on the fly (table log of the fixed-point D)" if derived else "stored",
            "kcap512_sweeps_per_s": kcap512,
            "label_changes_last_sweep": stats["n_changes"], "K_final": stats["K"],
            "moving_regime": moving,
            "reference_default_options": defaults,
            "other_configs": other,
            "incremental_mode_sweeps_per_s_rank0": inc_sweeps_per_s,
            "coclustering_allreduce_ms": allreduce_ms, "coclustering_merge_path": merge_path, "coclustering_diag_ok": diag_ok,
            # Roofline of the dominant kernel — physical figures only.  achieved = the bytes this kernel reads per launch ÷ its mean
            # launch duration; frac = achieved ÷ the 8 TB/s HBM peak (<= 1 by construction, reproducible from the rocprofv3 kernel
            # stats under profiles/); `bound` names what limits the kernel; `served_from` the memory level its bytes come from.
            "roofline": {"kernel": kernel_name, "bound": bound, "served_from": served,
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBPS) if achieved else None,
                         "bytes_read_per_launch": read_bytes, "avg_launch_ms": bulk_avg_ms, "launches": bulk_launches, "timed_every_nth_launch": time_every,
                         "per_sweep_period": {"ms": period_s * 1e3, "GBps": read_bytes / period_s / 1e9, "frac": read_bytes / period_s / 1e9 / HBM_PEAK_GBPS,
                                              "note": "the same bytes ÷ ms_per_step (launches of consecutive sweeps overlap on two streams, so a launch lasts longer than a period)"},
                         "traffic": traffic_rec["k_bulk_hbm_bytes_per_launch"] if traffic_rec else None, "traffic_source": traffic_source,
                         "traffic_note": "bytes at the L2's memory side per launch (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, separate passes, collected offline with this command); Infinity-Cache hits are counted there",
                         "compute": compute,
                         "measured_streaming_read_GBps": ceiling, "frac_of_measured_streaming_read": (achieved / ceiling) if achieved and ceiling else None,
                         # SURVEY.md §8(d)'s algorithmic pricing of the same launches (n²·sizeof per sweep with logD derived, 2·n²·sizeof stored):
                         # what a kernel that read every entry of the matrices would have to stream to keep up — NOT a bandwidth
                         "equivalent_dataflow_GBps": (survey_bytes / (bulk_avg_ms * 1e-3) / 1e9) if bulk_launches else None,
                         "equivalent_dataflow_bytes_per_sweep": survey_bytes,
                         "equivalent_dataflow_GBps_per_sweep_period": survey_bytes / period_s / 1e9,
                         "reference_dataflow_bytes_per_sweep": full_bytes,
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
