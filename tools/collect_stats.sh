#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command minus its host-only legs (run on the GPU box):
#   bash tools/collect_stats.sh <tag> ["extra bench flags"]
# -> gpurun_out/<tag>_kernel_stats.md (per-kernel table) + gpurun_out/<tag>_bench.json (bench line of the same run)
set -u
TAG=${1:-x}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$PWD
d=$R/gpurun_out/stats_$TAG
rm -rf $d
( cd /tmp && timeout -k 5 420 rocprofv3 --kernel-trace --stats -d $d -o x -- python $R/bench.py --no-cpu --no-parity --no-sharded ${2:-} > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err )
db=$(find $d -name "*results.db" | head -1)
python tools/rocpd_stats.py $db gpurun_out/${TAG}_kernel_stats.md > /dev/null
rm -rf $d
tail -2 gpurun_out/${TAG}_bench.err
