#!/bin/bash
# How long may a workgroup of the f32 vector scan hold its CU?  (round 6)  Experiment builds with VS_GRID_MULT = 1 (the
# persistent grid of rounds 2-5: 512 workgroups for the whole chunk), 8 and the product's 32; vector callers alone (what the pass costs) and
# hybrid callers (what a lexical launch waits for while a pass runs).  Run on the GPU box:  bash tools/probes/scan_grid.sh
set -u
cd "${GRAFT_REPO_ROOT:-.}"
for m in 1 8 32; do
  if [ $m = 32 ]; then lib=$PWD/seekstorm_amd/lib; else lib=$PWD/seekstorm_amd/lib_gm$m; fi
  [ -f $lib/libseekstorm_hip.so ] || SS_OUT_DIR=$lib SS_HIPCC_FLAGS="-DVS_GRID_MULT=${m}u" python -m seekstorm_amd.build > /dev/null 2>&1
  for leg in vector:64 hybrid:64,256; do
    echo "== VS_GRID_MULT=$m $leg"
    SEEKSTORM_HIP_LIB=$lib/libseekstorm_hip.so ONLY=$leg SSH_BENCH_HIST=1 SS_CO_TRACE=1 timeout 300 python tools/probes/concurrent_bench.py 3 2>&1 | grep -v amdgpu.ids
  done
done
