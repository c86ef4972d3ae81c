"""host-pointer latency of small lexical batches (the coalescer's batch sizes) against the partition count P of the pruned kernel:
    python tools/probes/small_batch_p.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import seekstorm_amd as S
from oracle import oracle as O
import bench
tl, th = bench.make_c2_queries(O, 1000)
sh = S.Shard(0)
sh.synth_lexical(O.LEX_SEED, 10_000_000, th, O.len_table())
sh.set_coalescing(0, 0, 0) if hasattr(sh, "set_coalescing") else None
for nq in (1, 8, 24, 41, 64, 100, 145, 256):
    q = sh.make_queries(tl[:nq], S.QueryType.Union)
    row = []
    for P in (0, 24, 32, 42, 48, 64, 96, 128, 149, 160, 240):
        if P:
            os.environ["SS_BM25_P"] = str(P)
        else:
            os.environ.pop("SS_BM25_P", None)
        for _ in range(20):
            sh.search_lexical_batch(q, 10, S.ResultType.Topk, reference_shortcuts=False)
        t0 = time.perf_counter()
        n = 300
        for _ in range(n):
            sh.search_lexical_batch(q, 10, S.ResultType.Topk, reference_shortcuts=False)
        row.append((P, (time.perf_counter() - t0) / n * 1e6))
    print("nq=%-4d " % nq + "  ".join("P=%s:%.0f" % (p if p else "def", us) for p, us in row), flush=True)
