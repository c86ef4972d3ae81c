"""Pruned (round 6: the wide instances of the probe kernel) vs exhaustive strategy for unions of 5..8 lists on the C2 corpus (10 M docs)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import seekstorm_amd as S
from seekstorm_amd import _native as N
from oracle import oracle as O

sh = S.Shard(0)
th = O.term_thresholds()
sh.synth_lexical(O.LEX_SEED, 10_000_000, th, O.len_table())
rng = np.random.default_rng(1)
df = np.array(sh.posting_count(np.arange(4096)), np.float64) / 1e7
bands = [np.nonzero((df >= a) & (df < b))[0] for a, b in ((0.005, 0.02), (0.02, 0.05), (0.05, 0.15), (0.002, 0.01), (0.01, 0.04), (0.03, 0.1), (0.1, 0.3), (0.001, 0.005))]
for nt in (5, 6, 7, 8):
    tl = [[int(rng.choice(bands[j])) for j in range(nt)] for _ in range(500)]
    q = sh.make_queries(tl, S.QueryType.Union)
    res = {}
    for name, strat in (("exhaustive", N.BM25_EXHAUSTIVE), ("auto", N.BM25_AUTO)):
        sh.set_strategy(strat)
        for rt in (S.ResultType.Topk, S.ResultType.TopkCount):
            r = sh.search_lexical_batch(q, 10, rt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                r = sh.search_lexical_batch(q, 10, rt)
            dt = (time.perf_counter() - t0) / 5
            res[(name, int(rt))] = (r, dt)
    for rt in (1, 2):
        a, b = res[("exhaustive", rt)], res[("auto", rt)]
        same = all(np.array_equal(x, y) for x, y in zip(a[0][:3], b[0][:3])) and (rt == 1 or np.array_equal(a[0][3], b[0][3]))
        print(f"nt={nt} rt={rt}: exhaustive {500 / a[1] / 1e3:.1f} K q/s, auto {500 / b[1] / 1e3:.1f} K q/s, identical={same}")
