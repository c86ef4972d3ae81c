"""head-heavy query mix (terms drawn with P ~ 1 / rank over the lists sorted by df, bench.py's realistic_vocabulary without its rare terms)
against the pruned kernel's partition count: host-pointer batches of 1000, ms per call"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import seekstorm_amd as S
from oracle import oracle as O
import bench
tl, th = bench.make_c2_queries(O, 1000)
sh = S.Shard(0)
sh.synth_lexical(O.LEX_SEED, 10_000_000, th, O.len_table())
rng = np.random.default_rng(99)
df = np.array([int(x) for x in sh.posting_count(list(range(len(th))))], np.int64)
by_rank = np.argsort(-df, kind="stable")
pr = 1.0 / np.arange(1, len(th) + 1, dtype=np.float64); pr /= pr.sum()
heavy = [sorted({int(by_rank[r]) for r in row}) for row in rng.choice(len(th), size=(1000, 3), p=pr)]
for name, lists in (("C2", tl), ("head-heavy", heavy)):
    q = sh.make_queries(lists, S.QueryType.Union)
    mean_post = float(np.mean([sum(df[t] for t in l) for l in lists]))
    row = []
    for P in (0, 16, 24, 32, 48, 64):
        if P: os.environ["SS_BM25_P"] = str(P)
        else: os.environ.pop("SS_BM25_P", None)
        for _ in range(5): sh.search_lexical_batch(q, 10, S.ResultType.Topk, reference_shortcuts=False)
        t0 = time.perf_counter()
        for _ in range(40): sh.search_lexical_batch(q, 10, S.ResultType.Topk, reference_shortcuts=False)
        row.append((P, (time.perf_counter() - t0) / 40 * 1e3))
    print("%-10s mean postings per query %.2f M: " % (name, mean_post / 1e6) + "  ".join("P=%s:%.3f" % (p if p else "def", ms) for p, ms in row), flush=True)
