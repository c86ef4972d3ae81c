#!/bin/bash
# per-launch timeline of the i8 vector pass (kernel trace only): bash tools/trace_vec8.sh <tag>
set -u
TAG=${1:-x}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$PWD
d=$R/gpurun_out/trace_$TAG
rm -rf $d
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $d -o x -- python $R/bench.py --workload vec --no-cpu --steps 4 --warmup 1 > $R/gpurun_out/${TAG}_trace_bench.json 2> $R/gpurun_out/${TAG}_trace_bench.err )
db=$(find $d -name "*results.db" | head -1)
python - "$db" > gpurun_out/${TAG}_trace.txt <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = cur.execute(f"select {name}, start, end, grid_x from kernels order by start").fetchall() if "grid_x" in cols else \
       [r + (0,) for r in cur.execute(f"select {name}, start, end from kernels order by start").fetchall()]
# last i8 pass: the final run of vec8 / refine launches
idx = [i for i, r in enumerate(rows) if "vec8_scan" in r[0]]
last = idx[-7:] if len(idx) >= 7 else idx
t0 = rows[last[0]][1]
for i in range(last[0] - 2, min(len(rows), last[-1] + 3)):
    n, s, e, g = rows[i]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  grid {g:7d}  {n[:60]}")
PY
rm -rf $d
cat gpurun_out/${TAG}_trace.txt
