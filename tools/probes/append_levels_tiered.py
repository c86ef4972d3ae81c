"""Commits on an image WITH A SPARSE TIER at the C2 size: 10 M docs in 153 levels; per level the dense terms (512 of the bench's
lists) through ss_bm25_append_level and ~0.7 M postings of 1 M rare terms (skewed list sizes) through ss_bm25_append_sparse_level --
the tier ends at ~107 M postings, the size of bench.py's realistic_vocabulary leg.  Per commit: ms of either call.
    python tools/probes/append_levels_tiered.py [levels]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import seekstorm_amd as S
from oracle import oracle as O
from oracle import fullsize as F
import bench
n_docs = 10_000_000
n_levels = min(int(sys.argv[1]) if len(sys.argv) > 1 else 153, (n_docs + 65535) // 65536)
n_sp, per_level = 1_000_000, 720_000
th_all = O.term_thresholds()
tl, _ = bench.make_c2_queries(O, 1000)
terms = sorted({t for q in tl for t in q})[:512]
t0 = time.time()
dl = O.lex_doclen(n_docs)
with ThreadPoolExecutor(F.host_threads(32)) as ex:
    parts = list(ex.map(lambda t: O.lex_term(int(t), th_all[int(t)], n_docs), terms))
print("host generation %.1f s, %d dense postings" % (time.time() - t0, sum(len(d) for d, _ in parts)), flush=True)
bounds = [np.searchsorted(d, np.arange(154, dtype=np.uint64) * 65536) for d, _ in parts]
rng = np.random.default_rng(3)
inc = S.Shard(0)
ms_d, ms_s = [], []
t_wall = time.time()
for lv in range(n_levels):
    lo, hi = lv * 65536, min(n_docs, (lv + 1) * 65536)
    offs = np.zeros(len(terms) + 1, np.uint64)
    offs[1:] = np.cumsum([int(b[lv + 1] - b[lv]) for b in bounds])
    docs = np.concatenate([parts[i][0][bounds[i][lv]:bounds[i][lv + 1]] for i in range(len(terms))]).astype(np.uint32)
    tfs = np.concatenate([parts[i][1][bounds[i][lv]:bounds[i][lv + 1]] for i in range(len(terms))]).astype(np.uint16)
    # the level's rare-term postings: (list, doc) pairs, low list ids more often than high ones
    key = np.unique(((n_sp * rng.random(per_level) ** 2).astype(np.uint64) << np.uint64(16)) | rng.integers(0, hi - lo, per_level).astype(np.uint64))
    s_list = (key >> np.uint64(16)).astype(np.int64)
    s_docs = (lo + (key & np.uint64(0xFFFF))).astype(np.uint32)
    s_tfs = np.minimum(rng.geometric(0.6, len(key)), 40).astype(np.uint16)
    s_offs = np.zeros(n_sp + 1, np.uint64)
    s_offs[1:] = np.cumsum(np.bincount(s_list, minlength=n_sp))
    t1 = time.perf_counter()
    inc.append_level(lv, dl[lo:hi], offs, docs, tfs)
    t2 = time.perf_counter()
    inc.append_sparse_level(lv, s_offs, s_docs, s_tfs)
    t3 = time.perf_counter()
    ms_d.append((t2 - t1) * 1e3); ms_s.append((t3 - t2) * 1e3)
    if lv % 16 == 0 or lv == n_levels - 1:
        print("level %3d: dense %7d postings %.1f ms | sparse %7d postings %.1f ms, tier %s" % (lv, len(docs), ms_d[-1], len(key), ms_s[-1], inc.sparse_info()),
              flush=True)
print("%d commits: wall %.1f s (incl. host generation of the levels); dense append ms median %.1f last %.1f; sparse level ms median %.1f last %.1f max %.1f" %
      (n_levels, time.time() - t_wall, float(np.median(ms_d)), ms_d[-1], float(np.median(ms_s)), ms_s[-1], max(ms_s)), flush=True)
# a query over both tiers still answers
q = inc.make_queries([[0, 512 + 5], [512 + 1, 512 + 2, 3]], S.QueryType.Union)
print("totals of two unions over both tiers:", inc.search_lexical_batch(q, 10)[3].tolist(), flush=True)
