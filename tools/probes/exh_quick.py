"""quick A/B of the exhaustive 16-bit scan: kernel ms per 1000 C2 queries (k = 10) on the uniform and on the clustered corpus -- unions (Topk /
TopkCount), the bench's 2-term intersections (TopkCount), unions + one NOT term + 1 % tombstones (TopkCount) -- and a digest of the answers
(variants of the kernel must agree bit for bit)"""
import hashlib, os, sys
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
CORPORA = [c for c in os.environ.get("CORPORA", "uniform,clustered").split(",") if c]
for name, seed in (("uniform", O.LEX_SEED), ("clustered", O.LEX_SEED_CLUSTERED)):
    if name not in CORPORA:
        continue
    sh = S.Shard(0)
    sh.synth_lexical(seed, 10_000_000, th, O.len_table())
    sh.set_strategy(N.BM25_EXHAUSTIVE)
    nq, k = 1000, 10
    od = torch.empty((nq, k), dtype=torch.int32, device=dev); os_ = torch.empty((nq, k), dtype=torch.float32, device=dev)
    oc = torch.empty((nq,), dtype=torch.int32, device=dev); ot = torch.empty((nq,), dtype=torch.int64, device=dev)
    rng = np.random.default_rng(4321)
    ba, bb = bench.band_terms(th, 0.01, 0.05), bench.band_terms(th, 0.05, 0.20)
    and_l = [[int(rng.choice(ba)), int(rng.choice(bb))] for _ in range(nq)]
    rng = np.random.default_rng(2468)
    gone = np.unique(rng.integers(0, 10_000_000, 100_000, dtype=np.uint64))
    nband = bench.band_terms(th, 0.02, 0.05)
    nots = []
    for q_ in tl:
        t_ = int(rng.choice(nband))
        while t_ in q_:
            t_ = int(rng.choice(nband))
        nots.append([t_])
    legs = [("or3 Topk", sh.make_queries(tl, S.QueryType.Union), N.RT_TOPK, 2 | (3 << 8) | (3 << 16), False),
            ("or3 TopkCount", sh.make_queries(tl, S.QueryType.Union), N.RT_TOPKCOUNT, 2 | (3 << 8) | (3 << 16), False),
            ("and2 TopkCount", sh.make_queries(and_l, S.QueryType.Intersection), N.RT_TOPKCOUNT, 1 | 64 | (2 << 8) | (2 << 16), False),
            ("or3 NOT+tomb TopkCount", sh.make_queries(tl, S.QueryType.Union, nots), N.RT_TOPKCOUNT, 2 | (4 << 8) | (3 << 16) | (1 << 24), True)]
    for rn, q, rt, ops, dele in legs:
        sh.set_deleted(gone if dele else [])
        qd = torch.from_numpy(q.view(np.uint8).reshape(nq, -1).copy()).to(dev)

        def call():
            N.check(L.ss_bm25_search_dev(sh._h, nq, qd.data_ptr(), k, rt, ops, od.data_ptr(), os_.data_ptr(), oc.data_ptr(), ot.data_ptr(), None), "s")
        for _ in range(3):
            call()
        N.check(L.ss_shard_sync(sh._h), "sync")
        dig = hashlib.sha256(os_.cpu().numpy().tobytes() + (ot.cpu().numpy().tobytes() if rt != N.RT_TOPK else b"")).hexdigest()[:12]
        sh.profile(True); sh.profile_read(0, reset=True)
        for _ in range(int(os.environ.get("REPS", 30))):
            call()
        N.check(L.ss_shard_sync(sh._h), "sync")
        c, ms = sh.profile_read(0, reset=True)
        sh.profile(False)
        print(f"{name:9s} {rn:24s} {ms / max(c, 1):.3f} ms   {dig}", flush=True)
    sh.close()
