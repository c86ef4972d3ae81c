"""batch-64 AnnMode::Nprobe(16) on the C3 image (10 M x 768 f32, 256 clusters per level): latency of a batch, and the answers saved
for a comparison between SS_VEC_ANN_SPARSE=1 (VALU kernel over the interested queries of a tile) and =0 (MFMA kernel over the batch)"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import seekstorm_amd as S
from seekstorm_amd import _native as N
from oracle import oracle as O
rows, dim, B, k = int(os.environ.get("ROWS", 10_000_000)), 768, 64, 100
tag = os.environ.get("SS_VEC_ANN_SPARSE", "1")
dev = torch.device("cuda", 0)
sh = S.Shard(0)
sh.synth_vectors(O.VEC_SEED, rows, dim)
lc, cc = [], []
for l0 in range(0, rows, 65536):
    n_l = min(65536, rows - l0)
    cl = [256] * (n_l // 256) + ([n_l % 256] if n_l % 256 else [])
    lc.append(len(cl)); cc += cl
sh.set_clusters(lc, cc)
qv = torch.from_numpy(O.vec_gen(O.VECQ_SEED, 0, B, dim)).to(dev)
mode = S.AnnMode.Nprobe(16)._c()
L = S.lib()
doc = torch.empty((B, k), dtype=torch.int32, device=dev); sc = torch.empty((B, k), dtype=torch.float32, device=dev)
cnt = torch.empty((B,), dtype=torch.int32, device=dev); tot = torch.empty((B,), dtype=torch.int64, device=dev); ncl = torch.empty((B,), dtype=torch.int32, device=dev)
def step(n, m):
    N.check(L.ss_vec_search_ann_dev(sh._h, n, qv.data_ptr(), k, N.FLT_MIN_NEG, m, doc.data_ptr(), sc.data_ptr(), cnt.data_ptr(), tot.data_ptr(), ncl.data_ptr(), None), "ann")
def timed(n, m, reps):
    for _ in range(3): step(n, m)
    N.check(L.ss_shard_sync(sh._h), "sync")
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); step(n, m); N.check(L.ss_shard_sync(sh._h), "sync"); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))
t_ann = timed(B, C.addressof(mode), 30)
d, s_, c_ = doc.cpu().numpy().copy(), sc.cpu().numpy().copy(), cnt.cpu().numpy().copy()
t_all = timed(B, None, 15)
t_one = timed(1, C.addressof(mode), 50)
print("sparse=%s: batch-64 Nprobe(16) %.3f ms, AnnMode::All batch %.3f ms (ratio %.2f), single query Nprobe %.3f ms; counts ok %s" % (
    tag, t_ann, t_all, t_ann / t_all, t_one, bool(np.all(c_ == k))), flush=True)
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "ann_sparse_%s.npz" % tag)
other = os.path.join(os.path.dirname(out), "ann_sparse_%s.npz" % ("0" if tag == "1" else "1"))
np.savez(out, d=d, s=s_)
if os.path.exists(other):
    o = np.load(other)
    same_docs = sum(len(set(d[i]) ^ set(o["d"][i])) == 0 for i in range(B))
    print("vs the other kernel: %d of %d queries with identical doc sets, max |score diff| %.3g (rel %.3g)" % (
        same_docs, B, float(np.abs(np.sort(s_, 1) - np.sort(o["s"], 1)).max()), float((np.abs(np.sort(s_, 1) - np.sort(o["s"], 1)) / np.abs(np.sort(o["s"], 1))).max())), flush=True)
