#!/bin/bash
# VERDICT r2 #8a: the vector legs with SURVEY 8d's own row generator (Box-Muller N(0,1), device-only experiment) against the
# integer-exact one the parity tests can regenerate (uniform(-1, 1), then normalize_f32): same scan, same work.
cd "${GRAFT_REPO_ROOT:-.}"
for g in 0 1; do
  SS_VEC_SYNTH_BOXMULLER=$g python bench.py --workload vec --no-cpu --no-parity --no-concurrent --no-sharded --min-seconds 1.0 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
v=d.get('vector', d)
i8=v.get('i8',{})
print('rows boxmuller=$g: f32 batch-64 top-100 %.0f q/s (%.3f ms per call, %.1f TFLOP/s = %.3f of peak); i8 %.0f q/s; single query p50 %.3f ms; ann batch64 %.2f ms' % (
  v.get('value', d['value']), v.get('ms_per_call', 0), v['roofline']['achieved'], v['roofline']['frac'], i8.get('value', 0), v['latency_ms']['single_query_p50'], v.get('ann',{}).get('batch64_ms_p50', 0)))"
done
