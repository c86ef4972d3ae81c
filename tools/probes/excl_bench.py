"""exhaustive strategy under exclusions on C2 (bench.py's exhaustive_not_tombstones leg as a probe): C2 queries + one NOT term each,
1 % of the docs tombstoned; kernel time per 1000-query call for Topk / TopkCount / Count on the 16-bit tile and on the f32 tile, for the
partition counts given on the command line (SS_BM25_P), answers compared between the tiles.
    python tools/probes/excl_bench.py [P ...]"""
import os, sys, time
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
sh.set_deleted(np.unique(rng.integers(0, n_docs, n_docs // 100, dtype=np.uint64)))
nband = bench.band_terms(th, 0.02, 0.05)
nl = []
for t in tl:
    x = int(rng.choice(nband))
    while x in t:
        x = int(rng.choice(nband))
    nl.append([x])
q = sh.make_queries(tl, S.QueryType.Union, nl)
nq, k = len(q), 10
qd = torch.from_numpy(q.view(np.uint8).reshape(nq, -1).copy()).to(dev)
od = torch.empty((nq, k), dtype=torch.int32, device=dev); os_ = torch.empty((nq, k), dtype=torch.float32, device=dev)
oc = torch.empty((nq,), dtype=torch.int32, device=dev); ot = torch.empty((nq,), dtype=torch.int64, device=dev)
L = S.lib()
OPS = 2 | (4 << 8) | (3 << 16) | (1 << 24)
def call(rt):
    N.check(L.ss_bm25_search_dev(sh._h, nq, qd.data_ptr(), k, rt, OPS, od.data_ptr(), os_.data_ptr(), oc.data_ptr(), ot.data_ptr(), None), "search")
def timed(rt, n=30):
    for _ in range(3): call(rt)
    N.check(L.ss_shard_sync(sh._h), "sync")
    sh.profile(True); sh.profile_read(0, reset=True)
    for _ in range(n): call(rt)
    N.check(L.ss_shard_sync(sh._h), "sync")
    c, ms = sh.profile_read(0, reset=True)
    sh.profile(False)
    return ms / max(c, 1)
ref = {}
sh.set_strategy(N.BM25_EXHAUSTIVE_F32)
for name, rt in (("Topk", N.RT_TOPK), ("TopkCount", N.RT_TOPKCOUNT), ("Count", N.RT_COUNT)):
    ms = timed(rt, 8)
    ref[name] = (os_.cpu().numpy().copy(), ot.cpu().numpy().copy())
    print("f32 tile   %-9s kernel %.3f ms" % (name, ms), flush=True)
sh.set_strategy(N.BM25_EXHAUSTIVE)
for P in [int(x) for x in sys.argv[1:]] or [0]:
    if P:
        os.environ["SS_BM25_P"] = str(P)
    for name, rt in (("Topk", N.RT_TOPK), ("TopkCount", N.RT_TOPKCOUNT), ("Count", N.RT_COUNT)):
        ms = timed(rt)
        ok = (rt == N.RT_COUNT or np.array_equal(os_.cpu().numpy(), ref[name][0])) and (rt == N.RT_TOPK or np.array_equal(ot.cpu().numpy(), ref[name][1]))
        print("16-bit P=%-3s %-9s kernel %.3f ms   equal to the f32 tile: %s" % (P or "auto", name, ms, ok), flush=True)
