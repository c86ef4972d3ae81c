cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_small_batch.py tests/test_gpu_pruned.py tests/test_gpu_union_many.py -x -q > gpurun_out/r5_kthb2_tests.log 2>&1; tail -3 gpurun_out/r5_kthb2_tests.log
NQS=1,8,16,32,64 timeout 300 python tools/probes/small_fused.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_kthb2_small.log; tail -14 gpurun_out/r5_kthb2_small.log
LEX_ONLY=1 timeout 300 python tools/probes/concurrent_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_kthb2_conc.log; tail -8 gpurun_out/r5_kthb2_conc.log
B="--workload bm25 --quick --steps 10 --calls-per-step 50 --no-rationed --no-fields --no-vocab --no-clustered --no-real-format --no-commit --no-concurrent"
timeout 300 python bench.py $B > gpurun_out/r5_kthb2_base.out 2> gpurun_out/r5_kthb2_base.err
SEEKSTORM_HIP_LIB=$GRAFT_REPO_ROOT/seekstorm_amd/lib_exp1/libseekstorm_hip.so timeout 300 python bench.py $B > gpurun_out/r5_kthb2_exp.out 2> gpurun_out/r5_kthb2_exp.err
SEEKSTORM_HIP_LIB=$GRAFT_REPO_ROOT/seekstorm_amd/lib_exp1/libseekstorm_hip.so timeout 600 python -m pytest tests/test_gpu_pruned.py tests/test_gpu_parity.py -x -q > gpurun_out/r5_kthb2_exptests.log 2>&1; tail -3 gpurun_out/r5_kthb2_exptests.log
for f in gpurun_out/r5_kthb2_base.out gpurun_out/r5_kthb2_exp.out; do python - "$f" <<'P'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith('{"metric"')][-1])
print(sys.argv[1], d["value"], d["roofline"]["frac"], d["roofline"].get("avg_launch_ms"), d["roofline"]["pruned"], d.get("latency_ms"))
P
done
