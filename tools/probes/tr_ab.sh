#!/bin/bash
# A/B of the transposed register chunks (S16_TR=1, lib_tr) against the 16-byte-per-lane form (product build at the time: lib): C2 exhaustive
# scan (3 / 2 terms), quick legs with answer digests (variants must agree bit for bit), 16- / 8- / 32-term unions; alternated twice
cd "${GRAFT_REPO_ROOT:-.}"
A=${TR_A:-$PWD/seekstorm_amd/lib_tr/libseekstorm_hip.so}
for rep in 1 2; do
  for v in tr base; do
    if [ $v = tr ]; then export SEEKSTORM_HIP_LIB=$A; else export SEEKSTORM_HIP_LIB=$PWD/seekstorm_amd/lib_base/libseekstorm_hip.so; fi
    python tools/probes/exh_time.py $v 2>&1 | grep variant
    EXH_NT=2 python tools/probes/exh_time.py $v 2>&1 | grep variant
    for nt in 16 8 32; do echo -n "$v NT=$nt "; NT=$nt python tools/probes/union16_time.py 2>&1 | grep auto; done
  done
done
for v in tr base; do
  if [ $v = tr ]; then export SEEKSTORM_HIP_LIB=$A; else export SEEKSTORM_HIP_LIB=$PWD/seekstorm_amd/lib_base/libseekstorm_hip.so; fi
  echo "== $v"; CORPORA=uniform REPS=20 python tools/probes/exh_quick.py 2>&1 | grep "uniform"
done
