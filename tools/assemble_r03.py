"""Copies the round-3 evidence from gpurun_out/r03 (written by tools/prof_r03.sh on the GPU box) into profiles/r03 and derives
pmc_traffic_n8192_k_bulk_syml2_true_true.json (what bench.py reports as roofline.traffic)."""
import glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S, P = os.path.join(ROOT, "gpurun_out", "r03") + "/", os.path.join(ROOT, "profiles", "r03") + "/"
os.makedirs(P, exist_ok=True)
raw = json.load(open(S + "pmc_summary_raw.json"))
g = lambda k: raw[k][0]
K = "k_bulk_syml2<true, true>"
fetch_factor = (1 << 30) / (g("calib:calib_read:FETCH_SIZE") * 1024)
write_factor = (32 << 20) / (g("calib:calib_atomic:WRITE_SIZE") * 1024)
f = g(f"headline:{K}:FETCH_SIZE") * 1024
w = g(f"headline:{K}:WRITE_SIZE") * 1024
n = 8192
out = {"kernel": K, "n": n,
       "command": "RC_BENCH_NO_TIMING=1 ... rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 (tools/prof_r03.sh; one counter per pass)",
       "FETCH_SIZE_KiB_per_launch_raw": f / 1024, "WRITE_SIZE_KiB_per_launch_raw": w / 1024, "launches_averaged": raw[f"headline:{K}:FETCH_SIZE"][1],
       "calibration": {"tool": "tools/calib_fetch.hip under the same rocprofv3 --pmc passes", "read_kernel": "1 GiB of 16-byte-per-lane non-temporal loads",
                       "fetch_correction_factor_measured": fetch_factor, "atomic_kernel": "32 MiB of 64-bit no-return atomic adds", "write_correction_factor_measured": write_factor,
                       "note": "FETCH_SIZE under-reports a wide streaming read by the factor 2 the MI355X guide states (measured %.4f); WRITE_SIZE counts the 64-bit atomics exactly. "
                               "FETCH_SIZE is counted at the L2's memory-side requests: Infinity-Cache hits are included — the 201 MB this kernel reads (the upper triangle of the "
                               "48-bit packed D) fit the 256 MiB Infinity Cache, so most of these bytes do not come from HBM" % fetch_factor},
       "read_bytes_per_launch_l2_memory_side": 2 * f, "hbm_write_bytes_per_launch": w, "k_bulk_hbm_bytes_per_launch": 2 * f + w,
       "algorithmic_bytes_survey_8d": n * n * 8, "bytes_the_kernel_has_to_read": n * (n + 1) // 2 * 6,
       "k_resolve_stationary": {"FETCH_SIZE_KiB_raw": g("headline:k_resolve:FETCH_SIZE"), "WRITE_SIZE_KiB_raw": g("headline:k_resolve:WRITE_SIZE")},
       "k_resolve_moving": {"FETCH_SIZE_KiB_raw": g("moving:k_resolve:FETCH_SIZE"), "WRITE_SIZE_KiB_raw": g("moving:k_resolve:WRITE_SIZE")}}
json.dump(out, open(P + "pmc_traffic_n8192_k_bulk_syml2_true_true.json", "w"), indent=1)
print("traffic MB per launch: read (L2 memory side) %.1f written %.1f; calibration factors %.4f %.4f" % (2 * f / 1e6, w / 1e6, fetch_factor, write_factor))
cp = lambda a, b: shutil.copy(S + a, P + b)
for a, b in (("pmc_summary_raw.json", "pmc_summary_raw.json"), ("counters_headline.txt", "counters_headline_n8192.txt"), ("counters_moving.txt", "counters_moving_regime_n8192.txt"),
             ("bench_default.json", "bench_default.json"), ("bench_steps20.json", "bench_steps20.json"), ("bench_headline_under_rocprof.json", "bench_headline_under_rocprof.json"),
             ("moving_under_rocprof.json", "moving_rate_under_rocprof.json"), ("stats/b_kernel_stats.csv", "kernel_stats_headline.csv"),
             ("stats_moving/b_kernel_stats.csv", "kernel_stats_moving_regime.csv"), ("calib_FETCH_SIZE.txt", "calib_fetch_stdout.txt")):
    try: cp(a, b)
    except FileNotFoundError as e: print("missing", e)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for leg in ("headline", "moving"):
        fs = glob.glob(S + f"pmc_{leg}_{ctr}/**/*counter_collection.csv", recursive=True)
        if fs:
            rows = open(fs[0]).read().splitlines()
            keep = [rows[0]] + [r for r in rows[1:] if "k_bulk" in r or "k_resolve" in r][:400]      # (the full CSVs are tens of MB)
            open(P + f"pmc_{leg}_{ctr.lower()}_counter_collection_head.csv", "w").write("\n".join(keep) + "\n")
for fn in ("bench_default.json", "bench_steps20.json", "bench_headline_under_rocprof.json"):
    d = json.loads(open(S + fn).read().strip().splitlines()[-1]); r = d["roofline"]
    print(fn, "sweeps/s %.0f  kernel %.1f us  frac(8d) %.3f  frac_on_bytes_read %.3f" % (d["value"], r["avg_launch_ms"] * 1e3, r["frac"], r["frac_on_bytes_read"]),
          "moving %s" % (d.get("moving_regime") and round(d["moving_regime"]["sweeps_per_s"])), "defaults %s" % (d.get("reference_default_options") and round(d["reference_default_options"]["iterations_per_s"])))
