"""exhaustive strategy at k = 100 on C2: the 16-bit tile with two keys per lane (KPL = 2 instances) against the f32 tile it replaces
(SS_BM25_EXHAUSTIVE_F32); plain 3-term unions, and the same with one NOT term + 1 % tombstones; answers compared between the tiles."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import seekstorm_amd as S
from seekstorm_amd import _native as N
from oracle import oracle as O
import bench
dev = torch.device("cuda", 0)
sh = S.Shard(0)
tl, th = bench.make_c2_queries(O, 1000)
n_docs = 10_000_000
sh.synth_lexical(O.LEX_SEED, n_docs, th, O.len_table())
rng = np.random.default_rng(2468)
nband = bench.band_terms(th, 0.02, 0.05)
nl = []
for t in tl:
    x = int(rng.choice(nband))
    while x in t:
        x = int(rng.choice(nband))
    nl.append([x])
L = S.lib()
for k in (100, 128):
  for what in ("plain", "NOT + tombstones"):
    if what == "plain":
        sh.set_deleted([])
        q = sh.make_queries(tl, S.QueryType.Union); OPS = 2 | (3 << 8) | (3 << 16)
    else:
        sh.set_deleted(np.unique(rng.integers(0, n_docs, n_docs // 100, dtype=np.uint64)))
        q = sh.make_queries(tl, S.QueryType.Union, nl); OPS = 2 | (4 << 8) | (3 << 16) | (1 << 24)
    nq = len(q)
    qd = torch.from_numpy(q.view(np.uint8).reshape(nq, -1).copy()).to(dev)
    od = torch.empty((nq, k), dtype=torch.int32, device=dev); os_ = torch.empty((nq, k), dtype=torch.float32, device=dev)
    oc = torch.empty((nq,), dtype=torch.int32, device=dev); ot = torch.empty((nq,), dtype=torch.int64, device=dev)
    def call(rt):
        N.check(L.ss_bm25_search_dev(sh._h, nq, qd.data_ptr(), k, rt, OPS, od.data_ptr(), os_.data_ptr(), oc.data_ptr(), ot.data_ptr(), None), "search")
    def timed(rt, n=10):
        for _ in range(2): call(rt)
        N.check(L.ss_shard_sync(sh._h), "sync")
        sh.profile(True); sh.profile_read(0, reset=True)
        for _ in range(n): call(rt)
        N.check(L.ss_shard_sync(sh._h), "sync")
        c, ms = sh.profile_read(0, reset=True)
        sh.profile(False)
        return ms / max(c, 1)
    for name, rt in (("Topk", N.RT_TOPK), ("TopkCount", N.RT_TOPKCOUNT)):
        sh.set_strategy(N.BM25_EXHAUSTIVE_F32)
        ms_f = timed(rt, 5)
        ref = (od.cpu().numpy().copy(), os_.cpu().numpy().copy(), ot.cpu().numpy().copy())
        sh.set_strategy(N.BM25_EXHAUSTIVE)
        ms_s = timed(rt)
        ok = np.array_equal(os_.cpu().numpy(), ref[1]) and np.array_equal(od.cpu().numpy(), ref[0]) and (rt == N.RT_TOPK or np.array_equal(ot.cpu().numpy(), ref[2]))
        print(f"k={k} {what:17s} {name:9s}: f32 tile {ms_f:.3f} ms, 16-bit tile {ms_s:.3f} ms per 1000 queries; same answers: {ok}", flush=True)
