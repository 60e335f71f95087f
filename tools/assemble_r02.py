"""Copies the round-2 evidence from gpurun_out/r02 (written by tools/prof_r02.sh on the GPU box) into profiles/r02 and derives
pmc_traffic_n8192_k_bulk_syml_true.json (what bench.py reports as roofline.traffic)."""
import json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S, P = os.path.join(ROOT, "gpurun_out", "r02") + "/", os.path.join(ROOT, "profiles", "r02") + "/"
raw = json.load(open(S + "pmc_summary_raw.json"))
g = lambda k: raw[k][0]
fetch_factor = (1 << 30) / (g("calib_FETCH_SIZE:calib_read:FETCH_SIZE_per_launch") * 1024)
write_factor = (32 << 20) / (g("calib_WRITE_SIZE:calib_atomic:WRITE_SIZE_per_launch") * 1024)
f = g("pmc_FETCH_SIZE:k_bulk_syml<true>:FETCH_SIZE_per_launch") * 1024
w = g("pmc_WRITE_SIZE:k_bulk_syml<true>:WRITE_SIZE_per_launch") * 1024
out = {"kernel": "k_bulk_syml<true>", "n": 8192,
       "command": "RC_BENCH_NO_TIMING=1 RC_BENCH_NO_INCREMENTAL=1 RC_BENCH_NO_MOVING=1 RC_BENCH_NO_DEFAULTS=1 rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 (tools/prof_r02.sh; one counter per pass)",
       "FETCH_SIZE_KiB_per_launch_raw": f / 1024, "WRITE_SIZE_KiB_per_launch_raw": w / 1024,
       "launches_averaged": raw["pmc_FETCH_SIZE:k_bulk_syml<true>:FETCH_SIZE_per_launch"][1],
       "calibration": {"tool": "tools/calib_fetch.hip under the same rocprofv3 --pmc passes", "read_kernel": "1 GiB of 16-byte-per-lane non-temporal loads",
                       "read_FETCH_SIZE_KiB_raw": g("calib_FETCH_SIZE:calib_read:FETCH_SIZE_per_launch"), "fetch_correction_factor_measured": fetch_factor,
                       "atomic_kernel": "32 MiB of 64-bit no-return atomic adds", "atomic_WRITE_SIZE_KiB_raw": g("calib_WRITE_SIZE:calib_atomic:WRITE_SIZE_per_launch"),
                       "write_correction_factor_measured": write_factor, "atomic_FETCH_SIZE_KiB_raw": g("calib_FETCH_SIZE:calib_atomic:FETCH_SIZE_per_launch"),
                       "note": "FETCH_SIZE under-reports the streaming read by the factor 2 the MI355X guide states (measured %.4f); WRITE_SIZE counts the 64-bit atomics exactly; the read half of an L2 atomic does not appear in FETCH_SIZE" % fetch_factor},
       "hbm_read_bytes_per_launch": 2 * f, "hbm_write_bytes_per_launch": w, "k_bulk_hbm_bytes_per_launch": 2 * f + w,
       "algorithmic_bytes_survey_8d": 8192 * 8192 * 8, "bytes_the_kernel_has_to_read": 8192 * 8193 // 2 * 8,
       "k_resolve": {"FETCH_SIZE_KiB_raw": g("pmc_FETCH_SIZE:k_resolve:FETCH_SIZE_per_launch"), "WRITE_SIZE_KiB_raw": g("pmc_WRITE_SIZE:k_resolve:WRITE_SIZE_per_launch"),
                     "hbm_bytes_per_launch": 2 * 1024 * g("pmc_FETCH_SIZE:k_resolve:FETCH_SIZE_per_launch") + 1024 * g("pmc_WRITE_SIZE:k_resolve:WRITE_SIZE_per_launch")}}
json.dump(out, open(P + "pmc_traffic_n8192_k_bulk_syml_true.json", "w"), indent=1)
print("traffic MB per launch: read %.1f written %.1f" % (2 * f / 1e6, w / 1e6))
cp = lambda a, b: shutil.copy(S + a, P + b)
cp("pmc_summary_raw.json", "pmc_summary_raw.json"); cp("sq_summary.txt", "sq_counters_n8192.txt")
cp("bench.json", "bench_default.json"); cp("bench_steps20.json", "bench_steps20.json"); cp("bench_under_rocprof.json", "bench_headline_under_rocprof.json")
cp("stats/b_kernel_stats.csv", "bench_n8192_kernel_stats_headline.csv"); cp("stats_full/b_kernel_stats.csv", "bench_n8192_kernel_stats_full_default_run.csv")
cp("calib_FETCH_SIZE.txt", "calib_fetch_stdout.txt")
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    cp(f"pmc_{ctr}/p_counter_collection.csv", f"pmc_{ctr.lower()}_counter_collection.csv")
    cp(f"calib_{ctr}/c_counter_collection.csv", f"calib_{ctr.lower()}_counter_collection.csv")
for fn in ("bench.json", "bench_steps20.json", "bench_under_rocprof.json"):
    d = json.loads(open(S + fn).read().strip().splitlines()[-1]); r = d["roofline"]
    print(fn, "sweeps/s %.0f  kernel %.1f us  frac %.3f  frac_on_bytes_read %.3f  sweep-level %.3f" % (d["value"], r["avg_launch_ms"] * 1e3, r["frac"], r["frac_on_bytes_read"], d["sweep_frac_of_hbm_peak"]),
          "moving %s" % (d.get("moving_regime") and round(d["moving_regime"]["sweeps_per_s"])), "defaults %s" % (d.get("reference_default_options") and round(d["reference_default_options"]["iterations_per_s"])))
print(open(S + "stats/b_kernel_stats.csv").read().splitlines()[1][:120]); print(open(S + "stats/b_kernel_stats.csv").read().splitlines()[2][:120])
