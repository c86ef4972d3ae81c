#!/bin/bash
# timing of experiment builds of the exhaustive scan: bash tools/probes/exh_variants.sh "<variant dirs suffixes>"
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
for v in "" $1; do
  lib=$R/seekstorm_amd/lib${v:+_$v}/libseekstorm_hip.so
  SEEKSTORM_HIP_LIB=$lib python tools/probes/exh_time.py "$v"
done
