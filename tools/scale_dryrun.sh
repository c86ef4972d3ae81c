#!/bin/bash
# Multi-GPU readiness: runs the driver's own SCALE command -- bench.py --gpus N, one rank per GPU over RCCL -- for every N
# the node offers, short and self-verifying: every rank prints a SCALE_CHECK line (the ranks its communicator spans = ncclCommCount, the
# microseconds of ONE all-gather, a checksum of the merged answers) and rank 0 asserts that all communicators span N ranks and all ranks
# hold the same answers; the JSON line carries them under "scale_check".  Nothing here needs more than the GPUs that are present.
#   tools/scale_dryrun.sh [extra bench.py flags]
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
NG=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
echo "GPUs on this node: $NG"
for N in 1 2 4 8; do
  [ "$N" -le "$NG" ] || continue
  echo "=== bench.py --gpus $N"
  python bench.py --gpus "$N" --steps 3 --warmup 1 --calls-per-step 20 --scale-check --no-cpu --no-parity --no-rationed --no-fields --no-vocab \
      --no-concurrent --no-real-format --no-clustered --no-topk-count --min-seconds 0.5 "$@" 2> >(grep -E "SCALE_CHECK|Error|error|assert" >&2) \
    | python -c "import json,sys; l=json.loads(sys.stdin.readline()); print(json.dumps({k: l.get(k) for k in ('n_gpus','value','ms_per_step','scale_check')}))"
done
