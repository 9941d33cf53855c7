"""Per-kernel-class sums of one PMC counter from a rocprofv3 rocpd database, restricted to the steady decode
region (the last `--steps` sampler launches).  usage: python tools/rocpd_pmc.py <results.db> <COUNTER> <steps> <out.json>"""
import json
import re
import sqlite3
import sys


def main():
    dbp, counter, steps, outp = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    db = sqlite3.connect(dbp)
    rows = db.execute("select kernel_name, value, start, end from counters_collection where counter_name = ? order by start", (counter,)).fetchall()
    samp = [i for i, r in enumerate(rows) if "sampler_generate" in r[0]]
    if len(samp) < steps + 1:
        raise SystemExit(f"only {len(samp)} sampler launches found")
    i0, i1 = samp[-steps - 1] + 1, samp[-1] + 1          # kernels of the last `steps` decode steps
    per = {}
    for name, v, s, e in rows[i0:i1]:
        k = re.sub(r"\(.*", "", name).replace("void ", "")
        d = per.setdefault(k, [0, 0.0, 0])
        d[0] += 1; d[1] += float(v); d[2] += e - s
    out = {"counter": counter, "steps": steps, "launches": i1 - i0, "total_per_step": sum(v[1] for v in per.values()) / steps,
           "kernels": {k: {"calls_per_step": v[0] / steps, "value_per_call": v[1] / v[0], "avg_us": v[2] / v[0] / 1e3} for k, v in per.items()}}
    json.dump(out, open(outp, "w"), indent=1)
    print(json.dumps(out)[:400])


if __name__ == "__main__":
    main()
