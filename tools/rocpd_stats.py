#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace [--stats]) as a per-kernel table:
calls, total / average / min / max duration.  Usage: python tools/rocpd_stats.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name} order by sum(end-start) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, c, s, a, mn, mx in rows:
        lines.append(f"| `{n[:90]}` | {c} | {s / 1e6:.3f} | {a / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100.0 * s / tot:.1f} |")
    # the same, per LAUNCH SIZE, for the scan / probe kernels: a bench run launches them at 1000 queries (the measured legs) and at a few
    # queries (parity blocks, single-query latencies) -- the rows of the largest grid are the launches the roofline figures are about
    gcols = [c for c in cols if "grid" in c.lower() and "x" in c.lower()] or [c for c in cols if "grid" in c.lower()]
    if gcols:
        g = gcols[0]
        rows2 = cur.execute(f"select {name}, {g}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                            f"where {name} like '%scan%' or {name} like '%probe_kernel%' group by {name}, {g} having count(*) >= 3 "
                            f"order by {name}, {g} desc").fetchall()
        lines += ["", f"Per launch size (`{g}`, launches of one size grouped; >= 3 launches):", "",
                  "| kernel | grid | calls | total ms | avg us | min us | max us |", "|---|---|---|---|---|---|---|"]
        for n, gx, c, s_, a, mn, mx in rows2:
            lines.append(f"| `{n[:90]}` | {gx} | {c} | {s_ / 1e6:.3f} | {a / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} |")
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
