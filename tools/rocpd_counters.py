"""Per-kernel averages of whatever PMC counters a rocprofv3 rocpd database holds.  usage: python tools/rocpd_counters.py <results.db> <out.json> [name filter]"""
import json
import re
import sqlite3
import sys


def main():
    dbp, outp = sys.argv[1], sys.argv[2]
    filt = sys.argv[3] if len(sys.argv) > 3 else ""
    db = sqlite3.connect(dbp)
    rows = db.execute("select kernel_name, counter_name, value, start, end from counters_collection order by start").fetchall()
    per = {}
    for name, cname, v, s, e in rows:
        if filt and filt not in name:
            continue
        k = re.sub(r"\(.*", "", name).replace("void ", "")
        d = per.setdefault(k, {})
        c = d.setdefault(cname, [0, 0.0, 0])
        c[0] += 1; c[1] += float(v); c[2] += e - s
    out = {k: {c: {"calls": v[0], "per_call": v[1] / v[0], "avg_us": v[2] / v[0] / 1e3} for c, v in d.items()} for k, d in per.items()}
    json.dump(out, open(outp, "w"), indent=1)
    for k, d in out.items():
        print(k[:70])
        for c, v in d.items():
            print(f"    {c:28s} {v['per_call']:16.1f}   ({v['calls']} calls, {v['avg_us']:.1f} us)")


if __name__ == "__main__":
    main()
