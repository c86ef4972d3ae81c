#!/bin/bash
# Counter calibration on known byte counts (tools/probes/pmc_calib.hip): one rocprofv3 --pmc pass per counter group.
# Run on the GPU box:  bash tools/calib_pmc.sh <tag>   ->  gpurun_out/calib_<tag>.txt
set -u
TAG=${1:-x}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/calib_${TAG}.txt; : > $OUT
$R/tools/probes/pmc_calib.bin >> $OUT 2>&1
for g in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"; do
  d=$R/gpurun_out/calib_${TAG}_$(echo $g | tr ' ' '_')
  rm -rf $d
  ( cd /tmp && timeout 300 rocprofv3 --pmc $g --kernel-trace -d $d -o x -- $R/tools/probes/pmc_calib.bin > $d.log 2>&1 )
  db=$(find $d -name "*results.db" | head -1)
  echo "=== $g" >> $OUT
  if [ -n "$db" ]; then python tools/rocpd_pmc.py $db calib_ >> $OUT 2>&1; else tail -5 $d.log >> $OUT; fi
  rm -rf $d
done
cat $OUT
