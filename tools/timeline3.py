"""Steady-state window of a rocprofv3 kernel trace: start / end (µs, relative) of every sweep kernel in the window.
usage: python tools/timeline3.py <kernel_trace.csv> [first_index] [count]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in rows]
ev = [e for e in ev if e[0].startswith("k_bulk") or e[0] == "k_resolve" or e[0].startswith("k_sweep")]
ev.sort(key=lambda e: e[1])
i0 = int(sys.argv[2]) if len(sys.argv) > 2 else len(ev) // 2
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 16
t0 = ev[i0][1]
for n, s, e, q in ev[i0:i0 + cnt]:
    print(f"{n:22s} q{q:>3s} start {(s - t0) / 1e3:8.1f}  end {(e - t0) / 1e3:8.1f}  dur {(e - s) / 1e3:7.1f}")
res = [e for e in ev if e[0] == "k_resolve" or e[0].startswith("k_sweep")][len(ev) // 8:]
per = [(b[2] - a[2]) / 1e3 for a, b in zip(res[:-1], res[1:])]
per.sort()
print("k_resolve end-to-end period us: median %.1f  p10 %.1f  p90 %.1f" % (per[len(per) // 2], per[len(per) // 10], per[9 * len(per) // 10]))
