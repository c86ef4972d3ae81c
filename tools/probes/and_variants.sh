#!/bin/bash
# 2-term AND (exhaustive, 16-bit scan) under experiment builds and partition counts: bash tools/probes/and_variants.sh "<lib suffixes>" "<P values>"
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
for v in "" $1; do
  for P in ${2:-8 16}; do
    lib=$R/seekstorm_amd/lib${v:+_$v}/libseekstorm_hip.so
    echo "variant ${v:-base} P=$P: $(SS_BM25_P=$P SEEKSTORM_HIP_LIB=$lib python tools/probes/and_bench.py 2>/dev/null | grep '^exhaustive' | tr '\n' ' ')"
    echo "variant ${v:-base} P=$P union2: $(EXH_NT=2 SS_BM25_P=$P SEEKSTORM_HIP_LIB=$lib python tools/probes/exh_time.py 2>/dev/null | grep variant)"
  done
done
