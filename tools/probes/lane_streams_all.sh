#!/bin/bash
# lexical / vector / hybrid callers at the new default (two lanes from 32 callers, a stream per lane) against one lane
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do
for v in auto32 1; do
  echo "== SS_COALESCE_LANES=$v"
  SS_COALESCE_LANES=$v CB_T=8,64,256 python tools/probes/concurrent_bench.py 2 2>&1 | grep "^lexical\|^vector\|^hybrid"
done
done
