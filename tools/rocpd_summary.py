"""Summarises a rocprofv3 rocpd database (--kernel-trace) into a per-kernel stats table + inter-kernel gaps.
usage: python tools/rocpd_summary.py <results.db> [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    m = re.match(r"void skinny_gemm_kernel<(.*)>", name)
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        pro = {"0": "norm", "1": "attn", "2": "packed"}.get(a[4], a[4])
        epi = {"0": "qkv+rope+kv", "1": "resid", "2": "swiglu", "3": "logits"}.get(a[5], a[5])
        return f"skinny_gemm<{a[0]},nbg{a[1]},w{a[2]}x{a[3]},{pro}->{epi}>"
    return name.replace("void ", "")[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    stats = {}
    for name, s, e in rows:
        k = short(name)
        d = stats.setdefault(k, [0, 0, 1 << 62, 0])
        dur = e - s
        d[0] += 1; d[1] += dur; d[2] = min(d[2], dur); d[3] = max(d[3], dur)
    total = sum(v[1] for v in stats.values())
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {k} | {v[0]} | {v[1] / 1e6:.3f} | {v[1] / v[0] / 1e3:.2f} | {v[2] / 1e3:.2f} | {v[3] / 1e3:.2f} | {100 * v[1] / total:.1f} |")
    # gaps between consecutive kernels inside the dense decode region (last 60 % of dispatches)
    n = len(rows)
    sub = rows[int(n * 0.4):]
    gaps = [sub[i + 1][1] - sub[i][2] for i in range(len(sub) - 1)]
    gaps_s = sorted(gaps)
    busy = sum(e - s for _, s, e in sub)
    span = sub[-1][2] - sub[0][1]
    lines.append("")
    lines.append(f"dispatches: {n}; in the last 60 % of them: span {span / 1e6:.3f} ms, kernel busy {busy / 1e6:.3f} ms ({100 * busy / span:.1f} %), "
                 f"median gap {gaps_s[len(gaps_s) // 2] / 1e3:.2f} us, mean gap {sum(gaps) / len(gaps) / 1e3:.2f} us, p90 gap {gaps_s[int(len(gaps_s) * 0.9)] / 1e3:.2f} us")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
