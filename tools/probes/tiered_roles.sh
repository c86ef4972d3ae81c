#!/bin/bash
# Where a one-launch batch of tiered queries spends its time (round 6): experiment builds of the library with role 1 (dense partitions) or
# role 3 (sparse lists) switched off, the same probe of tools/real_format.py against each.  Run on the GPU box:
#   bash tools/probes/tiered_roles.sh "or3/tiered/64"
set -u
cd "${GRAFT_REPO_ROOT:-.}"
F=${1:-or3/tiered/64}
for v in 0 1 2 3; do
  SS_OUT_DIR=$PWD/seekstorm_amd/lib_exp$v SS_HIPCC_FLAGS="-DSM_DBG_SKIP=$v" python -m seekstorm_amd.build > /dev/null 2>&1
  echo "== SM_DBG_SKIP=$v (bit 0: no role 1, bit 1: no role 3)"
  SEEKSTORM_HIP_LIB=$PWD/seekstorm_amd/lib_exp$v/libseekstorm_hip.so timeout 600 python tools/real_format.py 1000000 1 --probe "$F" 2>&1 | grep "^and2/\|^or3/\|^phrase/"
done
