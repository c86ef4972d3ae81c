#!/bin/bash
# A/B of the transposed register chunks of the 16-bit scans (lib_tr: -DS16_TR=<mode>) against the 16-byte-per-lane form (lib_base: -DS16_TR=0):
# C2 exhaustive scan (3 / 2 terms), quick legs with answer digests (variants must agree bit for bit), 16-term unions; alternated twice.
#   S16_TR=1: every register chunk + scan16m; 2: the loads alone; 3: only the lists that are WRITTEN (all but a query's longest)
cd "${GRAFT_REPO_ROOT:-.}"
A=${TR_A:-$PWD/seekstorm_amd/lib_tr/libseekstorm_hip.so}
B=$PWD/seekstorm_amd/lib_base/libseekstorm_hip.so
for rep in 1 2; do
  for v in tr base; do
    if [ $v = tr ]; then export SEEKSTORM_HIP_LIB=$A; else export SEEKSTORM_HIP_LIB=$B; fi
    python tools/probes/exh_time.py $v 2>&1 | grep variant
    EXH_NT=2 python tools/probes/exh_time.py $v 2>&1 | grep variant
    if [ -z "${TR_NO16:-}" ]; then echo -n "$v NT=16 "; NT=16 python tools/probes/union16_time.py 2>&1 | grep auto; fi
  done
done
for v in tr base; do
  if [ $v = tr ]; then export SEEKSTORM_HIP_LIB=$A; else export SEEKSTORM_HIP_LIB=$B; fi
  echo "== $v"; CORPORA=${CORPORA:-uniform,clustered} REPS=20 python tools/probes/exh_quick.py 2>&1 | grep "uniform\|clustered"
done
