"""Summarise a rocprofv3 kernel trace: per-kernel durations, gaps between consecutive k_bulk launches and the
overlap of k_resolve with k_bulk.  usage: python tools/timeline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(r["Kernel_Name"].split("(")[0], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in rows]
ev.sort(key=lambda e: e[1])
bulk = [e for e in ev if e[0] == "k_bulk"]
res = [e for e in ev if e[0] == "k_resolve"]
bulk = bulk[len(bulk)//4:]; res = res[len(res)//4:]
import statistics as st
print("queues:", sorted(set(e[3] for e in bulk)), sorted(set(e[3] for e in res)))
print("k_bulk dur us   median %.1f mean %.1f" % (st.median([(e[2]-e[1])/1e3 for e in bulk]), st.mean([(e[2]-e[1])/1e3 for e in bulk])))
print("k_resolve dur us median %.1f mean %.1f" % (st.median([(e[2]-e[1])/1e3 for e in res]), st.mean([(e[2]-e[1])/1e3 for e in res])))
gaps = [(bulk[i+1][1]-bulk[i][2])/1e3 for i in range(len(bulk)-1)]
print("gap between consecutive k_bulk us: median %.1f mean %.1f max %.1f" % (st.median(gaps), st.mean(gaps), max(gaps)))
per = [(bulk[i+1][1]-bulk[i][1])/1e3 for i in range(len(bulk)-1)]
print("k_bulk start-to-start us: median %.1f mean %.1f" % (st.median(per), st.mean(per)))
# overlap of each resolve with any bulk
ov = []
for r in res:
    o = 0
    for b in bulk:
        o += max(0, min(r[2], b[2]) - max(r[1], b[1]))
    ov.append(o / max(1, r[2]-r[1]))
print("fraction of k_resolve time overlapped with a k_bulk: median %.2f" % st.median(ov))
