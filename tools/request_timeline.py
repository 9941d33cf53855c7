"""Where the GPU's time goes inside ONE 256-utterance request (tools/request_probe.py under rocprofv3 --kernel-trace): kernel time by class (decode chain,
persistent launch, sampler, prompt pass, vocoder, other), idle time between kernels, and what the GPU waited for (the kernel that follows each long gap).

usage: python tools/request_timeline.py <kernel_trace.csv> [--reps 3]      (the LAST of the probe's `reps` requests is analysed: the trace is cut at the largest gaps)"""
import argparse
import csv
import json
import re

import numpy as np


def short(n):
    return re.split(r"[<(]", re.sub(r"^void ", "", n))[0]


def klass(n):
    if n.startswith("persist_layer"): return "decode_persistent"
    if n.startswith("sampler") or n.startswith("embed") or n.startswith("fill_meta") or n.startswith("compact"): return "sampler_and_row_state"
    if n.startswith("skinny_gemm") or n.startswith("attn_decode") or n.startswith("attn_merge"): return "decode_chain"
    if n.startswith("prefill") or n.startswith("attn_prefill") or n.startswith("split_") or n.startswith("pack") or n.startswith("lora"): return "prompt_pass"
    if n.startswith("cnx") or n.startswith("conv") or n.startswith("gemm_split") or n.startswith("dwconv") or n.startswith("head_") or n.startswith("overlap") or n.startswith("dft") or n.startswith("vq") or n.startswith("istft"):
        return "vocoder"
    return "other:" + n


ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
rows = []
with open(a.csv) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
st = np.array([r[0] for r in rows], dtype=np.int64); en = np.array([r[1] for r in rows], dtype=np.int64)
gaps = st[1:] - np.maximum.accumulate(en)[:-1]
cut = np.sort(np.argsort(gaps)[-(a.reps):])                       # the largest gaps: model load | request 0 | request 1 | request 2
i0 = int(cut[-1]) + 1
sel = rows[i0:]
span = (sel[-1][1] - sel[0][0]) / 1e6
busy, by, gap_after, big = 0.0, {}, {}, []
cur_end = sel[0][0]
for (s, e, n) in sel:
    k = klass(n)
    by[k] = by.get(k, 0.0) + (e - s) / 1e6
    if s > cur_end:
        g = (s - cur_end) / 1e6
        gap_after[k] = gap_after.get(k, 0.0) + g
        if g > 0.05: big.append((round(g, 3), k))
    busy += max(0, e - max(s, cur_end)) / 1e6
    cur_end = max(cur_end, e)
big.sort(reverse=True)
print(json.dumps({"request_span_ms": round(span, 2), "kernels": len(sel), "gpu_busy_ms": round(busy, 2), "gpu_idle_ms": round(span - busy, 2),
                  "kernel_ms_by_class": {k: round(v, 2) for k, v in sorted(by.items(), key=lambda t: -t[1])},
                  "idle_ms_in_front_of_class": {k: round(v, 2) for k, v in sorted(gap_after.items(), key=lambda t: -t[1])},
                  "gaps_over_50us": len(big), "gaps_over_50us_ms": round(sum(g for g, _ in big), 2), "largest_gaps_ms": big[:12]}))
