"""What would a perfect threshold seed buy the pruned kernel?  The C2 batch is run twice: normally, and again with every query's shared
threshold LEFT at the value the first run ended on (SS_BM25_KEEP_TAU=1: bm_expand_kernel does not reset tau) -- the ceiling of any scheme
that establishes tau before the main pass (two-phase launches, seeds from list maxima).  Same answers either way."""
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
seed = O.LEX_SEED_CLUSTERED if len(sys.argv) > 1 and sys.argv[1] == "clustered" else O.LEX_SEED
sh.synth_lexical(seed, 10_000_000, th, O.len_table())
q = sh.make_queries(tl, S.QueryType.Union)
nq, k = len(q), 10
qd = torch.from_numpy(q.view(np.uint8).reshape(nq, -1).copy()).to(dev)
od = torch.empty((nq, k), dtype=torch.int32, device=dev); os_ = torch.empty((nq, k), dtype=torch.float32, device=dev)
oc = torch.empty((nq,), dtype=torch.int32, device=dev); ot = torch.empty((nq,), dtype=torch.int64, device=dev)
L = S.lib()
def call():
    N.check(L.ss_bm25_search_dev(sh._h, nq, qd.data_ptr(), k, N.RT_TOPK, 2 | (3 << 8), od.data_ptr(), os_.data_ptr(), oc.data_ptr(), ot.data_ptr(), None), "search")
def timed(n=30):
    sh.profile(True); sh.profile_read(0, reset=True)
    for _ in range(n): call()
    N.check(L.ss_shard_sync(sh._h), "sync")
    c, ms = sh.profile_read(0, reset=True)
    sh.profile(False)
    return ms / max(c, 1)
for _ in range(3): call()
N.check(L.ss_shard_sync(sh._h), "sync")
ref = os_.cpu().numpy().copy()
print("corpus %s: pruned kernel, thresholds from zero : %.3f ms per 1000 queries" % ("clustered" if seed != O.LEX_SEED else "uniform", timed()), flush=True)
os.environ["SS_BM25_KEEP_TAU"] = "1"
call(); N.check(L.ss_shard_sync(sh._h), "sync")
ms = timed()
print("            thresholds kept from the last run   : %.3f ms   (same answers: %s)" % (ms, np.array_equal(ref, os_.cpu().numpy())), flush=True)
