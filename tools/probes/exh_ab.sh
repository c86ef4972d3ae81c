#!/bin/bash
# A/B of the exhaustive union scan on C2: narrow groups (SS_BM25_NARROW=1, default) against the 16-byte-per-lane kernel (=0)
cd "${GRAFT_REPO_ROOT:-.}"
for v in 1 0 1 0; do
  SS_BM25_NARROW=$v python bench.py --workload bm25 --no-cpu --no-parity --no-topk-count --min-seconds 0.7 --steps 2 --calls-per-step 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
e=d['exhaustive']
print('narrow=$v exhaustive %.0f q/s  kernel %.3f ms  %.0f GB/s  frac %.3f | auto %.0f q/s' % (e['value'], e['roofline']['avg_launch_ms'], e['roofline']['achieved'], e['roofline']['frac'], d['value']))"
done
