"""Gaps between consecutive kernels of the steady decode loop from a rocprofv3 --kernel-trace CSV (start / end timestamps per dispatch):
    python tools/trace_gaps.py <dir with *_kernel_trace.csv> [last N dispatches]"""
import csv
import glob
import statistics
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 240
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
per = {}
for a, b in zip(rows, rows[1:]):
    ka = a["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    kb = b["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    gap = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    dur = (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
    per.setdefault((ka, kb), []).append((gap, dur))
for (ka, kb), v in sorted(per.items(), key=lambda kv: -len(kv[1])):
    if len(v) < 5:
        continue
    print(f"{ka:42s} -> {kb:42s} n={len(v):4d}  gap median {statistics.median(g for g, _ in v):7.2f} us   duration of the first median {statistics.median(x for _, x in v):8.2f} us")
