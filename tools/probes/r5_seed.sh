cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5_gputests1.log 2>&1; tail -5 gpurun_out/r5_gputests1.log
PBS= timeout 300 python tools/probes/small_fused.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_small4.log; tail -30 gpurun_out/r5_small4.log
B="--workload bm25 --quick --steps 10 --calls-per-step 50 --no-rationed --no-fields --no-vocab --no-clustered --no-real-format --no-commit --no-concurrent"
timeout 300 python bench.py $B > gpurun_out/r5_seed1.out 2> gpurun_out/r5_seed1.err
SS_BM25_SEED=0 timeout 300 python bench.py $B > gpurun_out/r5_seed0.out 2> gpurun_out/r5_seed0.err
for f in gpurun_out/r5_seed1.out gpurun_out/r5_seed0.out; do python - "$f" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["pruned"]["avg_launch_ms"], d["latency_ms"])
P
done
