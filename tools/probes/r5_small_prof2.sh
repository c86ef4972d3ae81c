# kernel durations of the one-launch path per batch size (rocprofv3 --kernel-trace; the table comes from the rocpd database)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for nq in 1 16 32 64; do
  d=gpurun_out/r5_small2_nq$nq
  rm -rf $d
  CHILD=1 OUT=/tmp/x.npz NQS=$nq timeout -k 5 60 rocprofv3 --kernel-trace --stats -d $d -o t -- python tools/probes/small_fused.py > gpurun_out/r5_small2_prof_nq$nq.log 2>&1 < /dev/null
  db=$(find $d -name "*results.db" 2>/dev/null | head -1)
  echo "== nq=$nq"
  if [ -n "$db" ]; then timeout 60 python tools/rocpd_stats.py $db gpurun_out/r5_small2_nq$nq.md > /dev/null 2>&1 < /dev/null; head -5 gpurun_out/r5_small2_nq$nq.md | cut -c1-220; fi
  rm -rf $d
done
