"""Gaps between consecutive kernels of the decode loop, from a rocprofv3 --kernel-trace CSV (kernel_trace.csv): where the step's time goes that no kernel owns.

usage: python tools/trace_gaps.py <kernel_trace.csv> [--main persist_layer_kernel] [--skip 200]
Prints one JSON object: per (previous kernel -> next kernel) pair the count, median / p10 / p90 gap in us, the median durations of the kernels, and the
period of the main kernel (start to start) = the step time under the profiler."""
import argparse
import csv
import json
import re
import sys

import numpy as np


def short(n):
    n = re.sub(r"^void ", "", n)
    return re.split(r"[<(]", n)[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--main", default="persist_layer_kernel")
    ap.add_argument("--skip", type=int, default=100, help="main-kernel launches skipped at the start (warm-up, graph capture)")
    args = ap.parse_args()
    rows = []
    with open(args.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    mains = [i for i, r in enumerate(rows) if r[2] == args.main]
    if len(mains) <= args.skip + 8:
        print(json.dumps({"error": "too few launches of the main kernel", "n": len(mains)})); return
    i0, i1 = mains[args.skip], mains[-1]
    pairs, durs = {}, {}
    for i in range(i0, i1):
        a, b = rows[i], rows[i + 1]
        pairs.setdefault(a[2] + " -> " + b[2], []).append((b[0] - a[1]) / 1e3)
        durs.setdefault(a[2], []).append((a[1] - a[0]) / 1e3)
    st = np.array([rows[i][0] for i in mains[args.skip:]], dtype=np.float64)
    per = np.diff(st) / 1e3
    q = lambda v: {"n": len(v), "p10": round(float(np.percentile(v, 10)), 2), "median": round(float(np.median(v)), 2), "p90": round(float(np.percentile(v, 90)), 2),
                   "mean": round(float(np.mean(v)), 2)}
    out = {"main": args.main, "period_us": q(per), "gaps_us": {k: q(v) for k, v in pairs.items()}, "durations_us": {k: q(v) for k, v in durs.items()}}
    # gaps by position inside a graph replay: every 4th boundary is a replay boundary when 4 steps are captured per graph
    k = args.main + " -> " + args.main
    seq = [(rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(i0, i1) if rows[i + 1][2] == args.main and rows[i][2] != args.main]
    if len(seq) >= 16:
        n = len(seq) // 4 * 4
        a = np.array(seq[:n]).reshape(-1, 4)
        out["gap_before_main_by_position_mod4_median_us"] = [round(float(x), 2) for x in np.median(a, axis=0)]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
