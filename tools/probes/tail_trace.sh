#!/bin/bash
# which batches make the vector / hybrid tail at T = 64: SS_CO_TRACE's per-batch outliers (ss_api.hip ss_shard_destroy) beside the call histogram
cd "${GRAFT_REPO_ROOT:-.}"
for r in $(seq 1 ${1:-3}); do
  for leg in vector:64 hybrid:64; do
    echo "== $leg run $r"
    SS_CO_TRACE=1 ONLY=$leg SSH_BENCH_HIST=1 timeout 300 python tools/probes/concurrent_bench.py 2 2>&1 | grep "^\[hist\]\|^vector\|^hybrid\|^\[co" | cut -c1-360
  done
done
