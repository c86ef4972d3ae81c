"""times the exhaustive 3-term union scan on C2 (1000 queries per call) without checking results (experiment builds)"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import seekstorm_amd as S
from seekstorm_amd import _native as N
from oracle import oracle as O
import bench
tag = sys.argv[1] if len(sys.argv) > 1 else ""
nt = int(os.environ.get("EXH_NT", "3"))
dev = torch.device("cuda", 0)
sh = S.Shard(0)
tl, th = bench.make_c2_queries(O, 1000)
if nt == 2:
    tl = [t[1:] for t in tl]
sh.synth_lexical(O.LEX_SEED, 10_000_000, th, O.len_table())
q = sh.make_queries(tl, S.QueryType.Union)
nq, k = len(q), 10
qd = torch.from_numpy(q.view(np.uint8).reshape(nq, -1).copy()).to(dev)
od = torch.empty((nq, k), dtype=torch.int32, device=dev); os_ = torch.empty((nq, k), dtype=torch.float32, device=dev)
oc = torch.empty((nq,), dtype=torch.int32, device=dev); ot = torch.empty((nq,), dtype=torch.int64, device=dev)
L = S.lib()
sh.set_strategy(N.BM25_EXHAUSTIVE)
def call():
    N.check(L.ss_bm25_search_dev(sh._h, nq, qd.data_ptr(), k, N.RT_TOPK, 2 | (nt << 8), od.data_ptr(), os_.data_ptr(), oc.data_ptr(), ot.data_ptr(), None), "search")
for _ in range(5):
    call()
N.check(L.ss_shard_sync(sh._h), "sync")
sh.profile(True); sh.profile_read(0, reset=True)
t0 = time.perf_counter()
for _ in range(200):
    call()
N.check(L.ss_shard_sync(sh._h), "sync")
dt = time.perf_counter() - t0
n, ms = sh.profile_read(0, reset=True)
print("variant %-5s narrow=%s nt=%d: %.3f ms per call, kernel %.3f ms" % (tag or "base", os.environ.get("SS_BM25_NARROW", "1"), nt, dt / 200 * 1e3, ms / max(n, 1)), flush=True)
