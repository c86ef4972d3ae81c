"""the pruned kernel at the headline's batch size against its partition count P: device-resident batches of nq queries through
ss_bm25_search_dev, wall time per call (kernel + merge), P from SS_BM25_P:   python tools/probes/pruned_p_sweep.py"""
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
k = 10
qd = torch.from_numpy(q.view(np.uint8).reshape(len(q), -1).copy()).to(dev)
od = torch.empty((1000, k), dtype=torch.int32, device=dev); os_ = torch.empty((1000, k), dtype=torch.float32, device=dev)
oc = torch.empty((1000,), dtype=torch.int32, device=dev); ot = torch.empty((1000,), dtype=torch.int64, device=dev)
L = S.lib()
for nq in (256, 500, 1000):
    row = []
    for P in (0, 6, 8, 12, 16, 24, 32, 48):
        if P: os.environ["SS_BM25_P"] = str(P)
        else: os.environ.pop("SS_BM25_P", None)
        def call():
            N.check(L.ss_bm25_search_dev(sh._h, nq, qd.data_ptr(), k, N.RT_TOPK, 2 | (3 << 8) | (3 << 16), od.data_ptr(), os_.data_ptr(), oc.data_ptr(), ot.data_ptr(), None), "s")
        for _ in range(5): call()
        N.check(L.ss_shard_sync(sh._h), "sync")
        t0 = time.perf_counter()
        for _ in range(100): call()
        N.check(L.ss_shard_sync(sh._h), "sync")
        row.append((P, (time.perf_counter() - t0) / 100 * 1e3))
    print("nq=%-5d " % nq + "  ".join("P=%s:%.3f" % (p if p else "def", ms) for p, ms in row), flush=True)
