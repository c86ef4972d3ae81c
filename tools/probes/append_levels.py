"""ss_bm25_append_level at the C2 size: the 10 M-doc / 4096-term corpus of the bench regenerated on the host (oracle generator), committed
level by level (153 levels of 65 536 docs); per commit: total ms, of which the device rebuild.  The final image's answers == the device-
generated one-shot image's (ss_bm25_synth) on the bench's 1000 C2 queries.
    python tools/probes/append_levels.py [n_terms]   (default 4096: ~70 s of host generation on 16 cores)"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import seekstorm_amd as S
from seekstorm_amd import _native as N
from oracle import oracle as O
from oracle import fullsize as F
import bench
n_docs = 10_000_000
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
th_all = O.term_thresholds()
tl, _ = bench.make_c2_queries(O, 1000)
terms = list(range(nt)) if nt >= 4096 else sorted({t for q in tl for t in q})[:nt]
t0 = time.time()
dl = O.lex_doclen(n_docs)
with ThreadPoolExecutor(F.host_threads(32)) as ex:
    parts = list(ex.map(lambda t: O.lex_term(int(t), th_all[int(t)], n_docs), terms))
print("host generation %.1f s, %d postings" % (time.time() - t0, sum(len(d) for d, _ in parts)), flush=True)
n_levels = (n_docs + 65535) // 65536
# per level CSR: cut every term's list at the level boundaries
bounds = [np.searchsorted(d, np.arange(n_levels + 1, dtype=np.uint64) * 65536) for d, _ in parts]
inc = S.Shard(0)
ms_all, ms_dev = [], []
t_wall = time.time()
for lv in range(n_levels):
    lo, hi = lv * 65536, min(n_docs, (lv + 1) * 65536)
    offs = np.zeros(len(terms) + 1, np.uint64)
    offs[1:] = np.cumsum([int(b[lv + 1] - b[lv]) for b in bounds])
    docs = np.concatenate([parts[i][0][bounds[i][lv]:bounds[i][lv + 1]] for i in range(len(terms))]).astype(np.uint32)
    tfs = np.concatenate([parts[i][1][bounds[i][lv]:bounds[i][lv + 1]] for i in range(len(terms))]).astype(np.uint16)
    if os.environ.get("POS"):  # with positions (POS=1): 0 .. tf - 1 of every posting -- the content is irrelevant to the commit's cost
        st = np.zeros(len(tfs) + 1, np.int64); st[1:] = np.cumsum(tfs.astype(np.int64))
        positions = (np.arange(st[-1], dtype=np.int64) - np.repeat(st[:-1], tfs.astype(np.int64))).astype(np.uint16)
        inc.append_level(lv, dl[lo:hi], offs, docs, tfs, positions=positions)
    else:
        inc.append_level(lv, dl[lo:hi], offs, docs, tfs)
    _, raw_b, a, b = inc.incremental_info()
    ms_all.append(a); ms_dev.append(b)
    if lv % 16 == 0 or lv == n_levels - 1:
        print("level %3d: postings %8d  append %.1f ms (device rebuild %.1f ms)  raw %.2f GB" % (lv, len(docs), a, b, raw_b / 1e9), flush=True)
print("153 commits: wall %.1f s (incl. host slicing); append ms: first %.1f, median %.1f, last %.1f, max %.1f; rebuild ms last %.1f" %
      (time.time() - t_wall, ms_all[0], float(np.median(ms_all)), ms_all[-1], max(ms_all), ms_dev[-1]), flush=True)
if nt >= 4096:
    ref = S.Shard(0)
    ref.synth_lexical(O.LEX_SEED, n_docs, th_all, O.len_table())
    for strat in (N.BM25_AUTO, N.BM25_EXHAUSTIVE):
        inc.set_strategy(strat); ref.set_strategy(strat)
        x = inc.search_lexical_batch(inc.make_queries(tl, S.QueryType.Union), 10)
        y = ref.search_lexical_batch(ref.make_queries(tl, S.QueryType.Union), 10)
        print("strategy %d: level-by-level image == one-shot device-generated image on 1000 C2 queries: %s" %
              (strat, all(np.array_equal(u, v) for u, v in zip(x, y))), flush=True)
