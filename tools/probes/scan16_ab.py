"""exhaustive strategy, SS_BM25_SCAN16 as set in the environment: kernel time per 1000-query call for several query shapes on C2"""
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
th = O.term_thresholds()
sh.synth_lexical(O.LEX_SEED, 10_000_000, th, O.len_table())
frac = th.astype(np.float64) / 2.0 ** 32
rng = np.random.default_rng(5)
def band(lo, hi): return np.nonzero((frac >= lo) & (frac < hi))[0]
shapes = {
    "3t C2 k10": ([[int(rng.choice(band(0.005, 0.02))), int(rng.choice(band(0.02, 0.05))), int(rng.choice(band(0.05, 0.15)))] for _ in range(1000)], 10),
    "3t C2 k100": ([[int(rng.choice(band(0.005, 0.02))), int(rng.choice(band(0.02, 0.05))), int(rng.choice(band(0.05, 0.15)))] for _ in range(1000)], 100),
    "2t k10": ([[int(rng.choice(band(0.02, 0.05))), int(rng.choice(band(0.05, 0.15)))] for _ in range(1000)], 10),
    "4t k10": ([[int(rng.choice(band(0.005, 0.02))), int(rng.choice(band(0.02, 0.05))), int(rng.choice(band(0.05, 0.15))), int(rng.choice(band(0.002, 0.005)))] for _ in range(1000)], 10),
    "5t k10": ([[int(rng.choice(band(0.005, 0.02))), int(rng.choice(band(0.02, 0.05))), int(rng.choice(band(0.05, 0.15))), int(rng.choice(band(0.002, 0.005))),
                 int(rng.choice(band(0.0005, 0.002)))] for _ in range(1000)], 10),
    "6t k10": ([[int(rng.choice(band(0.005, 0.02))), int(rng.choice(band(0.02, 0.05))), int(rng.choice(band(0.05, 0.15))), int(rng.choice(band(0.002, 0.005))),
                 int(rng.choice(band(0.0005, 0.002))), int(rng.choice(band(0.01, 0.03)))] for _ in range(1000)], 10),
    "1t k10": ([[int(rng.choice(band(0.02, 0.15)))] for _ in range(1000)], 10),
    "3t dense k10": ([[int(x) for x in rng.choice(band(0.15, 0.7), 3, replace=False)] for _ in range(1000)], 10),
    "2t sparse k10": ([[int(x) for x in rng.choice(band(0.0005, 0.005), 2, replace=False)] for _ in range(1000)], 10),
}
L = S.lib()
sh.set_strategy(N.BM25_EXHAUSTIVE)
for name, (tl, k) in shapes.items():
    tl = [t for t in tl if len(set(t)) == len(t)] + [tl[0]] * sum(len(set(t)) != len(t) for t in tl)
    tl = [t if len(set(t)) == len(t) else tl[0] for t in tl]
    if len(band(0.15, 0.7)) < 3 and "dense" in name:
        continue
    q = sh.make_queries(tl, S.QueryType.Union)
    nq = len(q)
    nt = max(len(t) for t in tl)
    qd = torch.from_numpy(q.view(np.uint8).reshape(nq, -1).copy()).to(dev)
    od = torch.empty((nq, k), dtype=torch.int32, device=dev); os_ = torch.empty((nq, k), dtype=torch.float32, device=dev)
    oc = torch.empty((nq,), dtype=torch.int32, device=dev); ot = torch.empty((nq,), dtype=torch.int64, device=dev)
    def call():
        N.check(L.ss_bm25_search_dev(sh._h, nq, qd.data_ptr(), k, N.RT_TOPK, 2 | (nt << 8), od.data_ptr(), os_.data_ptr(), oc.data_ptr(), ot.data_ptr(), None), "search")
    for _ in range(3):
        call()
    N.check(L.ss_shard_sync(sh._h), "sync")
    sh.profile(True); sh.profile_read(0, reset=True)
    for _ in range(30):
        call()
    N.check(L.ss_shard_sync(sh._h), "sync")
    n, ms = sh.profile_read(0, reset=True)
    print("scan16=%s %-16s kernel %.3f ms  checksum %d" % (os.environ.get("SS_BM25_SCAN16", "1"), name, ms / max(n, 1), int(od.sum().item())), flush=True)
