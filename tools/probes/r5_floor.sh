cd $GRAFT_REPO_ROOT
echo "== keep tau (ceiling of any threshold seeding), 10 M docs"; SS_BM25_KEEP_TAU=1 CHILD=1 OUT=/tmp/x.npz timeout 300 python tools/probes/small_fused.py 2>&1 | grep -v amdgpu.ids
echo "== floor: 100 K docs"; DOCS=100000 CHILD=1 OUT=/tmp/y.npz NQS=1,64 timeout 300 python tools/probes/small_fused.py 2>&1 | grep -v amdgpu.ids
echo "== 1 M docs"; DOCS=1000000 CHILD=1 OUT=/tmp/y.npz NQS=1,64 timeout 300 python tools/probes/small_fused.py 2>&1 | grep -v amdgpu.ids
