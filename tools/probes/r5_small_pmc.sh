# counters of the one-launch kernel at 64 queries per call (separate passes, kernel trace only)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for g in sqA sqB; do
  if [ $g = sqA ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"; else C="SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; fi
  d=gpurun_out/pmc_small_$g; rm -rf $d
  ( cd /tmp && CHILD=1 OUT=/tmp/x.npz NQS=${NQS:-64} timeout -k 5 200 rocprofv3 --pmc $C --kernel-trace -d $OLDPWD/$d -o x -- python $OLDPWD/tools/probes/small_fused.py > $OLDPWD/$d.log 2>&1 )
  db=$(find $d -name "*results.db" | head -1)
  echo "=== group $g"; python tools/rocpd_pmc.py $db bm25_small_kernel 2>&1
  rm -rf $d
done
