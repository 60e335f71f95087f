"""Copies the round-4 evidence from gpurun_out/r04 (written by tools/run_r04_evidence.sh / tools/prof_r04.sh on the GPU box) into
profiles/r04 and derives the files bench.py reads:
  pmc_traffic_n8192_k_bulk_syml2_true_true.json   bytes at the L2's memory side per launch (roofline.traffic)
  counters_n8192_k_bulk_syml2_true_true.json      instruction counters of the same kernel (roofline.compute)
  pmc_traffic_n32768_k_bulk_sym32.json            config 5: HBM bytes per launch of k_bulk_sym32 (other_configs…roofline.traffic)"""
import csv, glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S, P = os.path.join(ROOT, "gpurun_out", "r04") + "/", os.path.join(ROOT, "profiles", "r04") + "/"
os.makedirs(P, exist_ok=True)
raw = json.load(open(S + "pmc_summary_raw.json"))
g = lambda k: raw[k][0]
fetch_factor = (1 << 30) / (g("calib:calib_read:FETCH_SIZE") * 1024)
write_factor = (32 << 20) / (g("calib:calib_atomic:WRITE_SIZE") * 1024)
calib = {"tool": "tools/calib_fetch.hip under the same rocprofv3 --pmc passes", "read_kernel": "1 GiB of 16-byte-per-lane non-temporal loads",
         "fetch_correction_factor_measured": fetch_factor, "atomic_kernel": "32 MiB of 64-bit no-return atomic adds", "write_correction_factor_measured": write_factor}


def kernel_avg_us(stats_csv, name):
    for row in csv.DictReader(open(stats_csv)):
        if row["Name"].replace("void ", "").startswith(name):
            return float(row["AverageNs"]) / 1e3, int(row["Calls"])
    return None, 0


# ---- headline: k_bulk_syml2<true, true> at n = 8192
K = "k_bulk_syml2<true, true>"
n = 8192
f = g(f"headline:{K}:FETCH_SIZE") * 1024
w = g(f"headline:{K}:WRITE_SIZE") * 1024
avg_us, calls = kernel_avg_us(S + "stats/b_kernel_stats.csv", "k_bulk_syml2<true, true>")
json.dump({"kernel": K, "n": n,
           "command": "RC_BENCH_NO_TIMING=1 ... rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 (tools/prof_r04.sh; one counter per pass)",
           "FETCH_SIZE_KiB_per_launch_raw": f / 1024, "WRITE_SIZE_KiB_per_launch_raw": w / 1024, "launches_averaged": raw[f"headline:{K}:FETCH_SIZE"][1],
           "calibration": dict(calib, note="FETCH_SIZE under-reports a wide streaming read by the factor 2 the MI355X guide states (measured %.4f); WRITE_SIZE counts the 64-bit atomics exactly. "
                                           "FETCH_SIZE is counted at the L2's memory-side requests: Infinity-Cache hits are included — the 201 MB this kernel reads (the upper triangle of the 48-bit "
                                           "packed D) fit the 256 MiB Infinity Cache, so most of these bytes do not come from HBM" % fetch_factor),
           "read_bytes_per_launch_l2_memory_side": 2 * f, "hbm_write_bytes_per_launch": w, "k_bulk_hbm_bytes_per_launch": 2 * f + w,
           "algorithmic_bytes_survey_8d": n * n * 8, "bytes_the_kernel_has_to_read": n * (n + 1) // 2 * 6,
           "rocprofv3_kernel_stats_avg_us": avg_us, "rocprofv3_kernel_stats_calls": calls},
          open(P + "pmc_traffic_n8192_k_bulk_syml2_true_true.json", "w"), indent=1)
busy = g(f"headline:{K}:SQ_BUSY_CYCLES")
# SQ_BUSY_CYCLES sums the busy cycles of the 32 shader engines (8 XCDs x 4): ÷ 32 ÷ the launch duration = the shader clock during the launch
clock_ghz = busy / 32.0 / (avg_us * 1e3) if avg_us else None
counters = {"kernel": K, "n": n, "source": "rocprofv3 --pmc passes of tools/prof_r04.sh (profiles/r04/counters_headline_n8192.txt)",
            "SQ_INSTS_VALU": g(f"headline:{K}:SQ_INSTS_VALU"), "SQ_INSTS_SALU": g(f"headline:{K}:SQ_INSTS_SALU"), "SQ_INSTS_LDS": g(f"headline:{K}:SQ_INSTS_LDS"),
            "SQ_INSTS_VMEM_RD": g(f"headline:{K}:SQ_INSTS_VMEM_RD"), "SQ_INSTS_VMEM_WR": g(f"headline:{K}:SQ_INSTS_VMEM_WR"),
            "SQ_LDS_BANK_CONFLICT": g(f"headline:{K}:SQ_LDS_BANK_CONFLICT"), "SQ_LDS_IDX_ACTIVE": g(f"headline:{K}:SQ_LDS_IDX_ACTIVE"),
            "SQ_WAVES": g(f"headline:{K}:SQ_WAVES"), "SQ_WAVE_CYCLES": g(f"headline:{K}:SQ_WAVE_CYCLES"), "SQ_WAIT_INST_ANY": g(f"headline:{K}:SQ_WAIT_INST_ANY"),
            "SQ_BUSY_CYCLES": busy, "WRITE_SIZE_bytes": w,
            # tools/valu_rate.hip (profiles/r03/valu_issue_rates_gfx950.txt): with three or more waves per SIMD a 32-bit ALU instruction issues every 1.9 cycles,
            # 64-bit integer, FP64, DPP and conversion instructions every 3.0-3.9; the kernel's stream is mostly the latter (the table log in FP64, 64-bit adds)
            "cycles_per_valu_inst": 3.1, "cycles_per_valu_inst_source": "tools/valu_rate.hip at 3 waves per SIMD, FP64 / 64-bit integer classes (profiles/r03/valu_issue_rates_gfx950.txt)",
            "clock_ghz": clock_ghz, "clock_source": "SQ_BUSY_CYCLES / 32 shader engines / rocprofv3 average launch duration",
            "rocprofv3_kernel_stats_avg_us": avg_us}
json.dump(counters, open(P + "counters_n8192_k_bulk_syml2_true_true.json", "w"), indent=1)
print("headline: read %.1f MB written %.1f MB per launch (calibration %.4f / %.4f); VALU %.2f M, LDS conflict rate %.3f, clock %.2f GHz, rocprof avg %.1f us" %
      (2 * f / 1e6, w / 1e6, fetch_factor, write_factor, counters["SQ_INSTS_VALU"] / 1e6, counters["SQ_LDS_BANK_CONFLICT"] / counters["SQ_LDS_IDX_ACTIVE"], clock_ghz or 0, avg_us or 0))

# ---- config 5: k_bulk_sym32 at n = 32768 (HBM-resident)
K5 = "k_bulk_sym32<16>"
f5 = g(f"c5:{K5}:FETCH_SIZE") * 1024
w5 = g(f"c5:{K5}:WRITE_SIZE") * 1024
avg5, calls5 = kernel_avg_us(S + "stats_c5/b_kernel_stats.csv", "k_bulk_sym32")
n5 = 32768
need5 = 2 * (n5 * (n5 + 1) // 2) * 4
json.dump({"kernel": K5, "n": n5, "command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 tools/config5_rate.py 20 (tools/prof_r04.sh; one counter per pass)",
           "FETCH_SIZE_KiB_per_launch_raw": f5 / 1024, "WRITE_SIZE_KiB_per_launch_raw": w5 / 1024, "launches_averaged": raw[f"c5:{K5}:FETCH_SIZE"][1], "calibration": calib,
           "read_bytes_per_launch_l2_memory_side": 2 * f5, "hbm_write_bytes_per_launch": w5, "k_bulk_hbm_bytes_per_launch": 2 * f5 + w5,
           "bytes_the_kernel_has_to_read": need5, "read_over_need": 2 * f5 / need5,
           "rocprofv3_kernel_stats_avg_us": avg5, "rocprofv3_kernel_stats_calls": calls5,
           "frac_of_8TBps_on_bytes_it_has_to_read": (need5 / (avg5 * 1e-6) / 8e12) if avg5 else None,
           "note": "4.29 GB per launch: 17x the 256 MiB Infinity Cache — these bytes come from HBM"},
          open(P + "pmc_traffic_n32768_k_bulk_sym32.json", "w"), indent=1)
print("config 5: read %.2f GB (%.3f x what the kernel has to read) written %.1f MB per launch; rocprof avg %.1f us = %.3f of 8 TB/s" %
      (2 * f5 / 1e9, 2 * f5 / need5, w5 / 1e6, avg5 or 0, (need5 / (avg5 * 1e-6) / 8e12) if avg5 else 0))

cp = lambda a, b: shutil.copy(S + a, P + b)
for a, b in (("pmc_summary_raw.json", "pmc_summary_raw.json"), ("counters_headline.txt", "counters_headline_n8192.txt"), ("counters_moving.txt", "counters_moving_regime_n8192.txt"),
             ("counters_c5.txt", "counters_config5_n32768.txt"),
             ("bench_default.json", "bench_default.json"), ("bench_steps20.json", "bench_steps20.json"), ("bench_headline_under_rocprof.json", "bench_headline_under_rocprof.json"),
             ("moving_under_rocprof.json", "moving_rate_under_rocprof.json"), ("config5_under_rocprof.json", "config5_rate_under_rocprof.json"),
             ("stats/b_kernel_stats.csv", "kernel_stats_headline.csv"), ("stats_moving/b_kernel_stats.csv", "kernel_stats_moving_regime.csv"),
             ("stats_c5/b_kernel_stats.csv", "kernel_stats_config5.csv"), ("calib_FETCH_SIZE.txt", "calib_fetch_stdout.txt"), ("pytest_gpu.txt", "pytest_gpu.txt"),
             ("verify.log", "fuzz_parity_run.log")):
    try: cp(a, b)
    except FileNotFoundError as e: print("missing", e)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for leg in ("headline", "moving", "c5"):
        fs = glob.glob(S + f"pmc_{leg}_{ctr}/**/*counter_collection.csv", recursive=True)
        if fs:
            rows = open(fs[0]).read().splitlines()
            keep = [rows[0]] + [r for r in rows[1:] if "k_bulk" in r or "k_resolve" in r][:400]      # (the full CSVs are tens of MB)
            open(P + f"pmc_{leg}_{ctr.lower()}_counter_collection_head.csv", "w").write("\n".join(keep) + "\n")
for fn in ("bench_default.json", "bench_steps20.json", "bench_headline_under_rocprof.json"):
    try:
        d = json.loads(open(S + fn).read().strip().splitlines()[-1]); r = d["roofline"]
        print(fn, "sweeps/s %.0f  kernel %.1f us  frac %.3f (per period %.3f)  equivalent dataflow %.0f GB/s" % (d["value"], r["avg_launch_ms"] * 1e3, r["frac"], r["per_sweep_period"]["frac"], r["equivalent_dataflow_GBps"]),
              "moving %s" % (d.get("moving_regime") and round(d["moving_regime"]["sweeps_per_s"])), "defaults %s" % (d.get("reference_default_options") and round(d["reference_default_options"]["iterations_per_s"])))
    except Exception as e:   # noqa: BLE001
        print(fn, "unreadable:", e)
