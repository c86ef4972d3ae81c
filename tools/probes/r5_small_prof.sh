cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
LEX_ONLY=1 timeout 300 python tools/probes/concurrent_bench.py 2 > gpurun_out/r5_conc1.log 2>&1
for nq in 1 64; do
  CHILD=1 OUT=/tmp/x.npz NQS=$nq timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r5_small_nq$nq -o t -- python tools/probes/small_fused.py > gpurun_out/r5_small_prof_nq$nq.log 2>&1
done
grep -v amdgpu.ids gpurun_out/r5_conc1.log | tail; find gpurun_out/r5_small_nq1 gpurun_out/r5_small_nq64 -name "*kernel_stats.csv" | while read f; do echo $f; head -6 $f | cut -c1-260; done
