"""quick A/B of the exhaustive 16-bit scan: kernel ms per 1000 C2 queries (Topk / TopkCount, k = 10) on the uniform and on the clustered corpus"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import seekstorm_amd as S
from seekstorm_amd import _native as N
from oracle import oracle as O
import bench
dev = torch.device("cuda", 0)
tl, th = bench.make_c2_queries(O, 1000)
L = S.lib()
for name, seed in (("uniform", O.LEX_SEED), ("clustered", O.LEX_SEED_CLUSTERED)):
    sh = S.Shard(0)
    sh.synth_lexical(seed, 10_000_000, th, O.len_table())
    sh.set_strategy(N.BM25_EXHAUSTIVE)
    q = sh.make_queries(tl, S.QueryType.Union)
    nq, k = len(q), 10
    qd = torch.from_numpy(q.view(np.uint8).reshape(nq, -1).copy()).to(dev)
    od = torch.empty((nq, k), dtype=torch.int32, device=dev); os_ = torch.empty((nq, k), dtype=torch.float32, device=dev)
    oc = torch.empty((nq,), dtype=torch.int32, device=dev); ot = torch.empty((nq,), dtype=torch.int64, device=dev)
    for rn, rt in (("Topk", N.RT_TOPK), ("TopkCount", N.RT_TOPKCOUNT)):
        def call():
            N.check(L.ss_bm25_search_dev(sh._h, nq, qd.data_ptr(), k, rt, 2 | (3 << 8) | (3 << 16), od.data_ptr(), os_.data_ptr(), oc.data_ptr(), ot.data_ptr(), None), "s")
        for _ in range(3): call()
        N.check(L.ss_shard_sync(sh._h), "sync")
        sh.profile(True); sh.profile_read(0, reset=True)
        for _ in range(30): call()
        N.check(L.ss_shard_sync(sh._h), "sync")
        c, ms = sh.profile_read(0, reset=True)
        sh.profile(False)
        print(f"{name:9s} {rn:9s} {ms / max(c, 1):.3f} ms", flush=True)
    sh.close()
