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
        msteps = max(20, min(args.steps, 200))
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
        def default_options_leg(Dx, Px, labels, burn, mode="full"):
            cd = rc.Context(Dx, device=dev0, kcap=kcap, storage_bits=BITS)
            cd.set_params(**Px); cd.set_state(labels); cd.cocluster_reset()
            cd.set_mode(mode)                             # "full": the library default (as `value`); "incremental": what runsampler and the Julia glue set
            cd.attach_host_matrices(Dx)                   # the proposals' restricted scans read the host matrix (logD derived by the library)
            cd.run_chain(burn, 0, 10, 5, 1, 1, r, p, 1.0)             # warm-up (worker threads, pinned buffers, caches; burn-in on the moving data)
            its = 1000
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
            defaults["with_accepted_proposals"] = dict(legs, iterations=3000, data="paper dataset 1 (n = 100), random initial labels, splitmerge='intended'",
                                                       same_chain=legs["speculative"]["splitmerge_acceptances"] == legs["synchronous"]["splitmerge_acceptances"]
                                                       and legs["speculative"]["K_final"] == legs["synchronous"]["K_final"])
        except Exception as e:   # noqa: BLE001
            defaults["with_accepted_proposals"] = {"error": str(e)}
    del Dm

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
        # correction applied): collected OFFLINE with this same command (tools/prof_r03.sh) and committed — not measured in
        # this run; `traffic_source` names the file
        traffic = traffic_source = None
        tag = kernel_name.replace("<", "_").replace(">", "").replace(", ", "_").replace(" ", "")
        for rnd in ("r03", "r02", "r01"):
            for name in (f"pmc_traffic_n{n}_{tag}.json", f"pmc_traffic_n{n}_{kernel_family}.json"):
                f = os.path.join(ROOT, "profiles", rnd, name)
                if traffic is None and os.path.exists(f) and BITS == 64 and (name.endswith(f"{tag}.json") or not derived):
                    traffic = json.load(open(f))["k_bulk_hbm_bytes_per_launch"]
                    traffic_source = f"profiles/{rnd}/{name} (rocprofv3 --pmc, collected offline with the same command)"
        ceiling = None
        try:   # the fraction is also reported against a streaming-read ceiling measured on this box, now
            ceiling = rc.measure_read_ceiling(dev0, 2048, 5)
        except Exception:   # noqa: BLE001
            pass
        period_s = dt / args.steps
        out = {
            "metric": "Gibbs sweeps/sec (n×n distM)", "value": value, "unit": "sweeps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": f"i{BITS} fixed-point storage, i64 exact sums, f64 scores",
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
                       "settle_ms": settle_ms, "sweeps_per_s_before_settle": (world * args.steps / unsettled_t) if unsettled_t else None,
                       "settle_note": "untimed sweeps of the same workload before the warm-up steps (device clocks of a running chain; RC_BENCH_SETTLE_MS=0 disables)"},
            "logD": "derived on the fly (table log of the fixed-point D)" if derived else "stored",
            "kcap512_sweeps_per_s": kcap512,
            "sweep_GBps_algorithmic": value / world * survey_bytes / 1e9,          # whole sweep (not just the kernel) at §8(d) bytes
            "sweep_frac_of_hbm_peak": value / world * survey_bytes / 1e9 / HBM_PEAK_GBPS,
            "sweep_GBps_on_bytes_read": value / world * alg_bytes / 1e9,
            "sweep_frac_of_hbm_peak_on_bytes_read": value / world * alg_bytes / 1e9 / HBM_PEAK_GBPS,   # physical: bytes the kernel reads per sweep PERIOD
            "sweep_GBps_vs_reference_dataflow": value / world * full_bytes / 1e9,  # 2·n²·sizeof per sweep, what the reference reads
            "label_changes_last_sweep": stats["n_changes"], "K_final": stats["K"],
            "moving_regime": moving,
            "reference_default_options": defaults,
            "incremental_mode_sweeps_per_s_rank0": inc_sweeps_per_s,
            "coclustering_allreduce_ms": allreduce_ms, "coclustering_merge_path": merge_path, "coclustering_diag_ok": diag_ok,
            # roofline of the dominant kernel.  `frac_on_bytes_read` is the PHYSICAL figure (bytes the kernel has to read — the upper
            # triangle of D — ÷ its mean launch duration ÷ peak); `achieved` / `frac` use the algorithmic bytes SURVEY.md §8(d)
            # prescribes (see survey_bytes above), which this design undercuts by symmetry; `traffic` is the HBM bytes per launch
            # measured with the PMC counters.
            "roofline": {"kernel": kernel_name, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "measured_streaming_read_GBps": ceiling, "frac_of_measured_streaming_read": (achieved / ceiling) if achieved and ceiling else None,
                         "frac_on_bytes_read_of_measured_streaming_read": (achieved_read / ceiling) if achieved_read and ceiling else None,
                         "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic, "traffic_source": traffic_source,
                         "pricing": "frac_on_bytes_read (physical): the bytes this kernel has to read (the upper triangle only) ÷ mean launch duration; "
                                    "achieved / frac: SURVEY.md §8(d) algorithmic bytes per sweep (n²·sizeof when logD is derived on the fly, "
                                    "2·n²·sizeof when stored) ÷ the same duration; *_per_sweep_period: the same bytes ÷ the sweep period "
                                    "(launches of consecutive sweeps overlap on two streams, so a launch lasts longer than a period)",
                         "timed_every_nth_launch": time_every,
                         "avg_launch_ms": bulk_avg_ms, "launches": bulk_launches,
                         "algorithmic_bytes_per_launch": survey_bytes,
                         "bytes_read_by_kernel_per_launch": alg_bytes,
                         "achieved_on_bytes_read": achieved_read,
                         "frac_on_bytes_read": (achieved_read / HBM_PEAK_GBPS) if achieved_read else None,
                         "frac_on_bytes_read_per_sweep_period": alg_bytes / period_s / 1e9 / HBM_PEAK_GBPS,
                         "frac_per_sweep_period": survey_bytes / period_s / 1e9 / HBM_PEAK_GBPS,
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
