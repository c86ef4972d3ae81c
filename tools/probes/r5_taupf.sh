# staged probe kernel with / without the threshold prefetch (lib_exp1 = -DPB_STAGED_TAUPF=1), and the one-launch path after it
cd $GRAFT_REPO_ROOT
PBS= timeout 300 python tools/probes/small_fused.py > gpurun_out/r5_small3.log 2>&1
LEX_ONLY=1 timeout 300 python tools/probes/concurrent_bench.py 2 > gpurun_out/r5_conc2.log 2>&1
timeout 300 python bench.py --workload bm25 --quick --steps 10 --calls-per-step 50 --no-rationed --no-fields --no-vocab --no-clustered --no-real-format --no-commit --no-concurrent > gpurun_out/r5_taupf0.out 2> gpurun_out/r5_taupf0.err
SEEKSTORM_HIP_LIB=$GRAFT_REPO_ROOT/seekstorm_amd/lib_exp1/libseekstorm_hip.so timeout 300 python bench.py --workload bm25 --quick --steps 10 --calls-per-step 50 --no-rationed --no-fields --no-vocab --no-clustered --no-real-format --no-commit --no-concurrent > gpurun_out/r5_taupf1.out 2> gpurun_out/r5_taupf1.err
grep -v amdgpu.ids gpurun_out/r5_small3.log | tail -32; grep -v amdgpu.ids gpurun_out/r5_conc2.log | tail -4
for f in gpurun_out/r5_taupf0.out gpurun_out/r5_taupf1.out; do python - "$f" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["roofline"]["pruned"], d["latency_ms"])
P
done
