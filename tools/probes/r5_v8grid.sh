cd $GRAFT_REPO_ROOT
for g in 768 256 384 320 192 128 768 256; do
  echo "== SS_VEC8_GRID=$g"
  SS_VEC8_GRID=$g ANN_PREC=i8 ANN_MODES=all timeout 120 python tools/probes/ann_latency.py 2>&1 | grep -v amdgpu.ids | tail -3
done
