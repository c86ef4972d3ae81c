cd $GRAFT_REPO_ROOT
for it in 512 1024 2048; do echo "== ITEMS $it (G = 4)"; SS_BM25_SMALL_ITEMS=$it CHILD=1 OUT=/tmp/d$it.npz NQS=1,8,32,64 timeout 300 python tools/probes/small_fused.py 2>&1 | grep -v amdgpu.ids; done
echo "== ITEMS 1024 (G = 8)"; SEEKSTORM_HIP_LIB=$GRAFT_REPO_ROOT/seekstorm_amd/lib_exp1/libseekstorm_hip.so SS_BM25_SMALL_ITEMS=1024 CHILD=1 OUT=/tmp/g8.npz NQS=32,64 timeout 300 python tools/probes/small_fused.py 2>&1 | grep -v amdgpu.ids
echo "== staged"; SS_BM25_SMALL=0 CHILD=1 OUT=/tmp/st.npz NQS=1,8,32,64 timeout 300 python tools/probes/small_fused.py 2>&1 | grep -v amdgpu.ids | head -3
python - <<'P'
import numpy as np
a=np.load('/tmp/st.npz')
for n in ('d512','d1024','d2048','g8'):
    b=np.load(f'/tmp/{n}.npz'); bad=[k for k in b.files if not np.array_equal(a[k],b[k])]
    print(n, "vs staged:", "IDENTICAL" if not bad else bad[:5])
P
LEX_ONLY=1 timeout 300 python tools/probes/concurrent_bench.py 2 2>&1 | grep -v amdgpu.ids
