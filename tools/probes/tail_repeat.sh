#!/bin/bash
# the p99 / p50 of the T = 64 / 256 vector and hybrid callers over REPEATED 2-second runs (what bench.py times): one slow batch of 64 in a
# run of 12 K calls is its p99.   bash tools/probes/tail_repeat.sh [runs]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
for r in $(seq 1 ${1:-4}); do
  for leg in vector:64 hybrid:64; do
    ONLY=$leg SSH_BENCH_HIST=1 timeout 300 python tools/probes/concurrent_bench.py 2 2>&1 | grep "^\[hist\]\|^vector\|^hybrid" | cut -c1-330
  done
done
