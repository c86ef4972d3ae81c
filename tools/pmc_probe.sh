#!/bin/bash
# PMC counters of an arbitrary command, one rocprofv3 --pmc pass per counter group (never combined with API tracing).
#   bash tools/pmc_probe.sh <tag> <kernel-name-substring> <command...>   ->  gpurun_out/pmcp_<tag>.txt
set -u
TAG=$1; PAT=$2; shift 2
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmcp_${TAG}.txt; : > $OUT
declare -A G
G[sqA]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
G[sqB]="SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
G[sqC]="SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"
G[fetch]="FETCH_SIZE GRBM_GUI_ACTIVE"
G[write]="WRITE_SIZE"
for g in ${PMC_GROUPS:-sqA sqB sqC fetch write}; do
  d=$R/gpurun_out/pmcp_${TAG}_${g}
  rm -rf $d
  ( cd /tmp && timeout 600 rocprofv3 --pmc ${G[$g]} --kernel-trace -d $d -o x -- "$@" > $d.log 2>&1 )
  db=$(find $d -name "*results.db" | head -1)
  echo "=== group $g" >> $OUT
  if [ -n "$db" ]; then python tools/rocpd_pmc.py $db "$PAT" >> $OUT 2>&1; else tail -5 $d.log >> $OUT; fi
  rm -rf $d
done
cat $OUT
