"""quick probe: 2-term AND top-10 (C1-shaped queries on the C2 corpus), pruned vs exhaustive"""
import sys, time, ctypes as C, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import seekstorm_amd as S
from seekstorm_amd import _native as N
from oracle import oracle as O
import bench
L = N.lib()
th = O.term_thresholds()
rng = np.random.default_rng(1)
a = bench.band_terms(th, 0.01, 0.05); b = bench.band_terms(th, 0.05, 0.20)
tl = [[int(rng.choice(a)), int(rng.choice(b))] for _ in range(1000)]
sh = S.Shard(0)
sh.synth_lexical(O.LEX_SEED, 10_000_000, th, O.len_table())
q = sh.make_queries(tl, S.QueryType.Intersection)
dev = torch.device("cuda", 0)
qd = torch.from_numpy(q.view(np.uint8).reshape(len(q), -1).copy()).to(dev)
k = 10
od = torch.empty((1000, k), dtype=torch.int32, device=dev); os_ = torch.empty((1000, k), dtype=torch.float32, device=dev)
oc = torch.empty(1000, dtype=torch.int32, device=dev); ot = torch.empty(1000, dtype=torch.int64, device=dev)
res = {}
for name, strat in (("exhaustive", N.BM25_EXHAUSTIVE), ("pruned", N.BM25_PRUNED)):
    sh.set_strategy(strat)
    for rt, rn in ((N.RT_TOPK, "Topk"), (N.RT_TOPKCOUNT, "TopkCount")):
        def step():
            N.check(L.ss_bm25_search_dev(sh._h, 1000, qd.data_ptr(), k, rt, 1 | 64 | (2 << 8) | (2 << 16), od.data_ptr(), os_.data_ptr(), oc.data_ptr(), ot.data_ptr(), None), "s")
        step(); sh_sync = L.ss_shard_sync(sh._h)
        t0 = time.perf_counter()
        for _ in range(10): step()
        L.ss_shard_sync(sh._h)
        dt = (time.perf_counter() - t0) / 10
        res[(name, rn)] = (od.cpu().numpy().copy(), os_.cpu().numpy().copy(), ot.cpu().numpy().copy())
        print(name, rn, "%.3f ms/1000 queries -> %.0f q/s" % (dt * 1e3, 1000 / dt))
for rn in ("Topk", "TopkCount"):
    e, p = res[("exhaustive", rn)], res[("pruned", rn)]
    print(rn, "identical:", np.array_equal(e[0], p[0]) and np.array_equal(e[1], p[1]), "totals equal:", np.array_equal(e[2], p[2]) if rn == "TopkCount" else "-")
