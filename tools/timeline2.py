"""Per-sweep timeline from a rocprofv3 kernel trace: durations of the row-reduction kernel and k_resolve, and the
gaps between consecutive kernels.  usage: python tools/timeline2.py <kernel_trace.csv>"""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(r["Kernel_Name"].split("(")[0], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
ev = [e for e in ev if e[0].startswith("k_bulk") or e[0] == "k_resolve" or e[0].startswith("void k_bulk")]
ev.sort(key=lambda e: e[1])
ev = ev[len(ev) // 3:]
dur = {}
for n, s, e in ev:
    dur.setdefault(n, []).append((e - s) / 1e3)
for n, v in dur.items():
    print(f"{n:40s} n={len(v):4d} median {st.median(v):7.1f} us  mean {st.mean(v):7.1f}")
gaps = {}
for a, b in zip(ev[:-1], ev[1:]):
    gaps.setdefault(a[0][:12] + "->" + b[0][:12], []).append((b[1] - a[2]) / 1e3)
for k, v in gaps.items():
    print(f"gap {k:30s} n={len(v):4d} median {st.median(v):7.1f} us")
per = [b[1] - a[1] for a, b in zip([e for e in ev if e[0] != "k_resolve"][:-1], [e for e in ev if e[0] != "k_resolve"][1:])]
print("bulk start-to-start median us:", st.median(per) / 1e3)
