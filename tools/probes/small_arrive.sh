#!/bin/bash
# The arrival of a workgroup at its query's counter in bm25_small_kernel: the product's relaxed atomic behind s_waitcnt against the memory
# model's own form, ACQ_REL at agent scope (-DSM_ARRIVE_ACQREL=1: one buffer_wbl2 + buffer_inv per WORKGROUP -- round 5 measured a fence per
# WAVE at 2.6 x).  End-to-end latency per call of 1 / 8 / 32 / 64 queries on the C2 image, answers compared with the staged pipeline's.
# Run on the GPU box:  bash tools/probes/small_arrive.sh      (VERDICT r5 "next" 3; the bare protocol: tools/probes/small_litmus.hip)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
for v in 0 1; do
  if [ $v = 0 ]; then lib=$PWD/seekstorm_amd/lib; else lib=$PWD/seekstorm_amd/lib_acq; fi
  [ -f $lib/libseekstorm_hip.so ] || SS_OUT_DIR=$lib SS_HIPCC_FLAGS="-DSM_ARRIVE_ACQREL=$v" python -m seekstorm_amd.build > /dev/null 2>&1
  echo "== SM_ARRIVE_ACQREL=$v"
  SEEKSTORM_HIP_LIB=$lib/libseekstorm_hip.so PBS="" timeout 900 python tools/probes/small_fused.py 2>&1 | grep -v "amdgpu.ids\|^rc 0"
done
