"""pruned top-10 on the skewed corpus of tools/probes/skew_corpus.py: per-partition block maxima (SS_BM25_SUBMAX=1) against
list-level maxima (=0); both checked against the exhaustive strategy"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import seekstorm_amd as S
from seekstorm_amd import _native as N
from oracle import oracle as O
import skew_corpus
n_docs = int(os.environ.get("SKEW_DOCS", 8_000_000))
RL = int(os.environ.get("SKEW_REGION_LOG2", 16))
NQ = int(os.environ.get("SKEW_NQ", 1000))
dl, offs, docs, tfs = skew_corpus.build(O, n_docs, [0.012, 0.035, 0.10, 0.02, 0.06, 0.15], region_log2=RL)
dev = torch.device("cuda", 0)
sh = S.Shard(0)
sh.upload_lexical(n_docs, dl, offs, docs, tfs)
rng = np.random.default_rng(1)
tl = [[int(x) for x in rng.choice(6, 3, replace=False)] for _ in range(NQ)]
q = sh.make_queries(tl, S.QueryType.Union)
nq, k = len(q), 10
qd = torch.from_numpy(q.view(np.uint8).reshape(nq, -1).copy()).to(dev)
od = torch.empty((nq, k), dtype=torch.int32, device=dev); os_ = torch.empty((nq, k), dtype=torch.float32, device=dev)
oc = torch.empty((nq,), dtype=torch.int32, device=dev); ot = torch.empty((nq,), dtype=torch.int64, device=dev)
L = S.lib()
def call():
    N.check(L.ss_bm25_search_dev(sh._h, nq, qd.data_ptr(), k, N.RT_TOPK, 2 | (3 << 8), od.data_ptr(), os_.data_ptr(), oc.data_ptr(), ot.data_ptr(), None), "search")
sh.set_strategy(N.BM25_EXHAUSTIVE); call(); N.check(L.ss_shard_sync(sh._h), "sync"); ref = os_.cpu().numpy().copy()
sh.set_strategy(N.BM25_AUTO); call(); N.check(L.ss_shard_sync(sh._h), "sync")
assert np.array_equal(ref, os_.cpu().numpy()), "pruned differs from exhaustive"
for _ in range(5):
    call()
N.check(L.ss_shard_sync(sh._h), "sync")
t0 = time.perf_counter()
for _ in range(200):
    call()
N.check(L.ss_shard_sync(sh._h), "sync")
dt = time.perf_counter() - t0
sh.set_strategy(N.BM25_EXHAUSTIVE)
t0 = time.perf_counter()
for _ in range(50):
    call()
N.check(L.ss_shard_sync(sh._h), "sync")
de = time.perf_counter() - t0
print("skewed corpus, %d docs, regions of 2^%d docs, batch %d, SS_BM25_SUBMAX=%s: pruned %.3f ms per batch (%.0f q/s); exhaustive %.3f ms" % (
    n_docs, RL, NQ, os.environ.get("SS_BM25_SUBMAX", "auto"), dt / 200 * 1e3, nq * 200 / dt, de / 50 * 1e3), flush=True)
