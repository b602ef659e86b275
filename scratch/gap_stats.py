"""Gaps between consecutive kernels of the stepping loop from a rocprofv3 --kernel-trace CSV: where the step's time goes
between launches.  usage: gap_stats.py <kernel_trace.csv>"""
import csv, sys, statistics as st
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("void coflux::", "").replace("coflux::", "")
    return n.split("(")[0][:44]
gaps = {}
for a, b in zip(rows, rows[1:]):
    k = (short(a["Kernel_Name"]), short(b["Kernel_Name"]))
    gaps.setdefault(k, []).append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:8]:
    print(f"{k[0]:>46s} -> {k[1]:<46s} n={len(v):5d} median gap {st.median(v):7.2f} us  p10 {sorted(v)[len(v)//10]:7.2f}  p90 {sorted(v)[9*len(v)//10]:7.2f}")
dur = {}
for r in rows:
    dur.setdefault(short(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:5]:
    print(f"{k:>46s} n={len(v):5d} median {st.median(v):8.2f} us")
