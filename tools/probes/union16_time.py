"""64 unions of 16 terms on the C2 corpus: ms per call of the many-list scan (AUTO) and of the f32 tile"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import seekstorm_amd as S
from seekstorm_amd import _native as N
from oracle import oracle as O
import bench
tl, th = bench.make_c2_queries(O, 1000)
sh = S.Shard(0)
sh.synth_lexical(O.LEX_SEED, int(os.environ.get("DOCS", 10_000_000)), th, O.len_table())
rng = np.random.default_rng(1357)
lo_b, hi_b = bench.band_terms(th, 0.005, 0.05), bench.band_terms(th, 0.05, 0.15)
nt = int(os.environ.get("NT", 16))
lists = [[int(x) for x in rng.choice(lo_b, nt * 3 // 4, replace=False)] + [int(x) for x in rng.choice(hi_b, nt - nt * 3 // 4, replace=False)] for _ in range(128)]
q = sh.make_queries(lists[:65], S.QueryType.Union)  # 65 queries: the staged pipeline (not the one-launch path)
for name, strat in (("auto", N.BM25_AUTO), ("f32", N.BM25_EXHAUSTIVE_F32)):
    sh.set_strategy(strat)
    for _ in range(5):
        r = sh.search_lexical_batch(q, 10, S.ResultType.Topk, reference_shortcuts=False)
    t0 = time.perf_counter()
    n = int(os.environ.get("N", 50))
    for _ in range(n):
        r = sh.search_lexical_batch(q, 10, S.ResultType.Topk, reference_shortcuts=False)
    print("%-5s %.3f ms per call of 65 queries x %d terms   top score %.4f" % (name, (time.perf_counter() - t0) / n * 1e3, nt, r[1][0][0]), flush=True)
sh.close()
