cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ann.py tests/test_gpu_euclid.py -x -q -k "vec or vector or ann or euclid" > gpurun_out/r5_vq_tests.log 2>&1; tail -3 gpurun_out/r5_vq_tests.log
B="--workload vec --quick --steps 10 --no-concurrent --no-sharded --no-real-format"
timeout 400 python bench.py $B > gpurun_out/r5_vq1.out 2> gpurun_out/r5_vq1.err
SS_VEC_SCAN_Q=0 timeout 400 python bench.py $B > gpurun_out/r5_vq0.out 2> gpurun_out/r5_vq0.err
for f in gpurun_out/r5_vq1.out gpurun_out/r5_vq0.out; do python - "$f" <<'P'
import json,sys
L=[l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith('{"metric"')]
if not L: print(sys.argv[1], "no line"); sys.exit()
d=json.loads(L[-1])
print(sys.argv[1], d["metric"], d["value"], d["ms_per_step"], d["roofline"])
P
done
