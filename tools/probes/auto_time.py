"""times the AUTO (pruned) 3-term union top-10 on C2 and checks it against the exhaustive strategy"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import seekstorm_amd as S
from seekstorm_amd import _native as N
from oracle import oracle as O
import bench
tag = sys.argv[1] if len(sys.argv) > 1 else ""
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
def call(n=nq):
    N.check(L.ss_bm25_search_dev(sh._h, n, qd.data_ptr(), k, N.RT_TOPK, 2 | (3 << 8), od.data_ptr(), os_.data_ptr(), oc.data_ptr(), ot.data_ptr(), None), "search")
sh.set_strategy(N.BM25_EXHAUSTIVE); call(); N.check(L.ss_shard_sync(sh._h), "sync"); ref = os_.cpu().numpy().copy()
sh.set_strategy(N.BM25_AUTO); call(); N.check(L.ss_shard_sync(sh._h), "sync")
assert np.array_equal(ref, os_.cpu().numpy()), "pruned differs from exhaustive"
for _ in range(5):
    call()
N.check(L.ss_shard_sync(sh._h), "sync")
sh.profile(True); sh.profile_read(0, reset=True)
t0 = time.perf_counter()
for _ in range(300):
    call()
N.check(L.ss_shard_sync(sh._h), "sync")
dt = time.perf_counter() - t0
n, ms = sh.profile_read(0, reset=True)
t0 = time.perf_counter()
for _ in range(300):
    call(1)
N.check(L.ss_shard_sync(sh._h), "sync")
d1 = time.perf_counter() - t0
import ctypes as C
hd = np.empty((nq, k), np.uint32); hs = np.empty((nq, k), np.float32); hc = np.empty(nq, np.uint32); ht = np.empty(nq, np.uint64)
def hcall():
    N.check(L.ss_bm25_search(sh._h, nq, q.ctypes.data_as(C.c_void_p), k, N.RT_TOPK, N.ptr(hd, N.u32p), N.ptr(hs, N.f32p), N.ptr(hc, N.u32p), N.ptr(ht, N.u64p)), "ss_bm25_search")
for _ in range(5):
    hcall()
assert np.array_equal(hs, ref), "host-pointer call differs"
lat = []
for _ in range(300):
    t0 = time.perf_counter(); hcall(); lat.append(time.perf_counter() - t0)
lat = np.sort(lat) * 1e3
print("variant %-8s host pointers in, answers on the host: p50 %.3f ms p99 %.3f ms (%.0f q/s at the mean)" % (tag or "base", lat[150], lat[297], nq / (lat.mean() * 1e-3)), flush=True)
print("variant %-8s submax=%s: %.3f ms per 1000-query call (%.0f q/s), probe kernel %.3f ms; single query %.3f ms" % (
    tag or "base", os.environ.get("SS_BM25_SUBMAX", "1"), dt / 300 * 1e3, nq * 300 / dt, ms / max(n, 1), d1 / 300 * 1e3), flush=True)
