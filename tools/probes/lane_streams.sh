#!/bin/bash
# the coalescer's lanes on streams of their own (ss_common.h lstream): lexical callers for lane thresholds (SS_COALESCE_LANES=autoN) and
# for the batch size from which a lane's launch goes to the shared stream (SS_LANE_SHARE_FROM, a getenv switch at the time, a compile-time constant since; 100000 = never)
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do
for v in "auto32 48" "auto32 64" "auto32 100000" "auto48 48" "auto96 48" "1 48"; do
  set -- $v
  echo "== SS_COALESCE_LANES=$1 SS_LANE_SHARE_FROM=$2"
  SS_COALESCE_LANES=$1 SS_LANE_SHARE_FROM=$2 LEX_ONLY=1 CB_T=${CB_T:-8,32,64,96,128,192,256,512} python tools/probes/concurrent_bench.py 2 2>&1 | grep "^lexical"
done
done
