cd $GRAFT_REPO_ROOT
echo "== G = 8 (product)"; CHILD=1 OUT=/tmp/a.npz timeout 300 python tools/probes/small_fused.py 2>&1 | grep -v amdgpu.ids
echo "== G = 4"; SEEKSTORM_HIP_LIB=$GRAFT_REPO_ROOT/seekstorm_amd/lib_exp1/libseekstorm_hip.so CHILD=1 OUT=/tmp/b.npz timeout 300 python tools/probes/small_fused.py 2>&1 | grep -v amdgpu.ids
python - <<'P'
import numpy as np
a=np.load('/tmp/a.npz'); b=np.load('/tmp/b.npz')
bad=[k for k in a.files if not np.array_equal(a[k],b[k])]
print("G8 vs G4:", "IDENTICAL" if not bad else bad[:5])
P
echo "== G = 8 concurrent"; LEX_ONLY=1 timeout 300 python tools/probes/concurrent_bench.py 2 2>&1 | grep -v amdgpu.ids
