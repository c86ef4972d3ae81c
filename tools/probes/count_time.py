"""exhaustive strategy on C2: kernel time per 1000-query call for ResultType Topk / TopkCount / Count (SS_BM25_SCAN16_COUNT=0: the
f32 kernel's count mode), totals compared with the AUTO strategy's (bit records)"""
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
sh.synth_lexical(O.LEX_SEED, 10_000_000, th, O.len_table())
q = sh.make_queries(tl, S.QueryType.Union)
nq, k = len(q), 10
qd = torch.from_numpy(q.view(np.uint8).reshape(nq, -1).copy()).to(dev)
od = torch.empty((nq, k), dtype=torch.int32, device=dev); os_ = torch.empty((nq, k), dtype=torch.float32, device=dev)
oc = torch.empty((nq,), dtype=torch.int32, device=dev); ot = torch.empty((nq,), dtype=torch.int64, device=dev)
L = S.lib()
def call(rt):
    N.check(L.ss_bm25_search_dev(sh._h, nq, qd.data_ptr(), k, rt, 2 | (3 << 8), od.data_ptr(), os_.data_ptr(), oc.data_ptr(), ot.data_ptr(), None), "search")
sh.set_strategy(N.BM25_AUTO); call(N.RT_TOPKCOUNT); N.check(L.ss_shard_sync(sh._h), "sync")
ref_t, ref_s = ot.cpu().numpy().copy(), os_.cpu().numpy().copy()
sh.set_strategy(N.BM25_EXHAUSTIVE)
for name, rt in (("Topk", N.RT_TOPK), ("TopkCount", N.RT_TOPKCOUNT), ("Count", N.RT_COUNT)):
    for _ in range(3): call(rt)
    N.check(L.ss_shard_sync(sh._h), "sync")
    ok = (rt == N.RT_TOPK or np.array_equal(ot.cpu().numpy(), ref_t)) and (rt == N.RT_COUNT or np.array_equal(os_.cpu().numpy(), ref_s))
    sh.profile(True); sh.profile_read(0, reset=True)
    t0 = time.perf_counter()
    for _ in range(40): call(rt)
    N.check(L.ss_shard_sync(sh._h), "sync")
    dt = time.perf_counter() - t0
    n, ms = sh.profile_read(0, reset=True)
    print("scan16_count=%s %-9s: %.3f ms per call, kernel %.3f ms, %.0f q/s, equal to AUTO: %s" % (os.environ.get("SS_BM25_SCAN16_COUNT", "1"), name, dt / 40 * 1e3, ms / max(n, 1), nq * 40 / dt, ok), flush=True)
