"""Per-kernel sums of the MFMA-busy / SQ-busy counters from a rocprofv3 rocpd database.
usage: python tools/rocpd_mfma.py <results.db> <out.json>   (counters collected: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES)"""
import json
import re
import sqlite3
import sys


def main():
    dbp, outp = sys.argv[1], sys.argv[2]
    db = sqlite3.connect(dbp)
    rows = db.execute("select kernel_name, counter_name, value, start, end from counters_collection order by start").fetchall()
    per = {}
    for name, cname, v, s, e in rows:
        k = re.sub(r"\(.*", "", name).replace("void ", "")
        d = per.setdefault(k, {"calls": {}, "sum": {}, "ns": 0})
        d["calls"][cname] = d["calls"].get(cname, 0) + 1
        d["sum"][cname] = d["sum"].get(cname, 0.0) + float(v)
        if cname == "SQ_BUSY_CYCLES":
            d["ns"] += e - s
    out = {}
    for k, d in per.items():
        calls = max(d["calls"].values())
        mf, bz = d["sum"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), d["sum"].get("SQ_BUSY_CYCLES", 0.0)
        if mf <= 0:
            continue
        out[k] = {"calls": calls, "avg_us": d["ns"] / max(calls, 1) / 1e3, "SQ_VALU_MFMA_BUSY_CYCLES_per_call": mf / calls,
                  "SQ_BUSY_CYCLES_per_call": bz / calls,
                  # MFMA pipes busy as a share of the chip's SIMD-cycles over the kernel's duration (1024 SIMDs x duration x 2.4 GHz upper bound)
                  "mfma_busy_share_of_simd_cycles_at_2p4GHz": mf / (1024 * (d["ns"] * 2.4)) if d["ns"] else None}
    json.dump({"note": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES (own pass, no other trace domains); "
                       "SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs (MI355X_MICROARCH.md)", "kernels": out}, open(outp, "w"), indent=1)
    print(json.dumps(out)[:600])


if __name__ == "__main__":
    main()
