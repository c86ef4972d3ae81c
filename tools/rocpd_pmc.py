#!/usr/bin/env python3
"""Per-kernel PMC totals from a rocprofv3 rocpd database collected with --pmc ... --kernel-trace.
Usage: python tools/rocpd_pmc.py <results.db> [kernel-name-substring]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    rows = cur.execute("select * from counters_collection").fetchall()
    ix = {c: i for i, c in enumerate(cols)}
    kn = "kernel_name" if "kernel_name" in ix else "name"
    agg = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    dur = defaultdict(float)
    for r in rows:
        k = r[ix[kn]]
        if pat and pat not in k:
            continue
        agg[k][r[ix["counter_name"]]] += float(r[ix["value"]])
        disp[k].add(r[ix["dispatch_id"]])
    for k in agg:
        print(f"## {k[:100]}  dispatches={len(disp[k])}")
        for c, v in sorted(agg[k].items()):
            print(f"  {c:28s} total={v:.4g}  per_dispatch={v / max(len(disp[k]), 1):.4g}")


if __name__ == "__main__":
    main()
