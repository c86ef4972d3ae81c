#!/bin/bash
# A/B of s16_qm's folded decode (S16_QM_FOLD=1, product) against the round-2 form (lib_base, -DS16_QM_FOLD=0): C2 exhaustive scan,
# 2-term unions, 16-term unions; alternated twice on the same box
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do
  for v in fold base; do
    if [ $v = base ]; then export SEEKSTORM_HIP_LIB=$PWD/seekstorm_amd/lib_base/libseekstorm_hip.so; else unset SEEKSTORM_HIP_LIB; fi
    python tools/probes/exh_time.py $v 2>&1 | grep variant
    EXH_NT=2 python tools/probes/exh_time.py $v 2>&1 | grep variant
    echo -n "$v "; python tools/probes/union16_time.py 2>&1 | grep auto
  done
done
