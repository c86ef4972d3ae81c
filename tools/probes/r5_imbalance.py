"""is the one-launch kernel's time at nq = 64 set by its heaviest query?  64 copies of ONE query (light ... heavy) against the mixed batch"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import seekstorm_amd as S
from oracle import oracle as O
import bench
tl, th = bench.make_c2_queries(O, 1000)
sh = S.Shard(0)
sh.synth_lexical(O.LEX_SEED, 10_000_000, th, O.len_table())
uniq = sorted({t for q in tl[:64] for t in q})
df = dict(zip(uniq, (int(x) for x in sh.posting_count(uniq))))
w = [sorted(df[t] for t in q) for q in tl[:64]]
est = np.array([d[0] + d[1] / 2 + d[2] / 8 for d in w])
order = np.argsort(est)


def t_of(lists, n=300):
    q = sh.make_queries(lists, S.QueryType.Union)
    for _ in range(30):
        sh.search_lexical_batch(q, 10, S.ResultType.Topk, reference_shortcuts=False)
    lat = []
    for _ in range(n):
        t0 = time.perf_counter()
        sh.search_lexical_batch(q, 10, S.ResultType.Topk, reference_shortcuts=False)
        lat.append((time.perf_counter() - t0) * 1e6)
    return float(np.median(lat))


print("mixed 64: %.1f us" % t_of(tl[:64]))
for pos in (0, 16, 32, 48, 63):
    i = int(order[pos])
    print("64 x query %2d (est rank %2d, dfs %s, est %.0f): %.1f us   alone: %.1f us" % (i, pos, w[i], est[i], t_of([tl[i]] * 64), t_of([tl[i]])))
sh.close()
