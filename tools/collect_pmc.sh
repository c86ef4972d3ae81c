#!/bin/bash
# Collects rocprofv3 PMC counters for the bench in separate passes (one counter group per run; never combined with
# sys/hip/hsa tracing).  Run on the GPU box:  bash tools/collect_pmc.sh <workload: bm25|vec|all> <tag>
# Output: gpurun_out/pmc_<tag>_<group>/ (rocpd databases) + gpurun_out/pmc_<tag>.txt (per-kernel totals).
set -u
WL=${1:-bm25}; TAG=${2:-x}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_${TAG}.txt; : > $OUT
declare -A G
G[sqA]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
G[sqB]="SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
G[sqC]="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_IFETCH GRBM_GUI_ACTIVE"
G[fetch]="FETCH_SIZE GRBM_GUI_ACTIVE"
G[write]="WRITE_SIZE"
for g in ${3:-sqA sqB sqC fetch write}; do
  d=gpurun_out/pmc_${TAG}_${g}
  rm -rf $d
  ( cd /tmp && timeout -k 5 240 rocprofv3 --pmc ${G[$g]} --kernel-trace -d $OLDPWD/$d -o x -- python $OLDPWD/bench.py --workload $WL --quick --no-topk-count --steps 4 --warmup 1 > $OLDPWD/$d.log 2>&1 )
  db=$(find $d -name "*results.db" | head -1)
  echo "=== group $g ($db)" >> $OUT
  python tools/rocpd_pmc.py $db >> $OUT 2>&1
done
tail -3 gpurun_out/pmc_${TAG}_sqA.log >> $OUT
# the rocpd databases are far too large to travel back from the GPU box (gpurun_out/ is merged only below 64 MiB): reduce them
# HERE to the summary + pmc_traffic.json, keep those in gpurun_out/ (copy them into profiles/ afterwards), drop the databases
if [ -z "${3:-}" ]; then
  python tools/pmc_summary.py gpurun_out $TAG > /dev/null 2>> $OUT
  cp profiles/${TAG}_pmc_summary.md profiles/pmc_traffic.json gpurun_out/ 2>> $OUT
  rm -rf gpurun_out/pmc_${TAG}_*/
fi
