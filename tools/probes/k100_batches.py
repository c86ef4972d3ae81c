"""C2 3-term unions at k = 100 (the one-launch path's KPL = 2 instances): ms per call for 1 / 8 / 16 / 32 / 64 queries per host-pointer call,
answers checked against the exhaustive strategy; SEEKSTORM_HIP_LIB selects the build"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import seekstorm_amd as S
from seekstorm_amd import _native as N
from oracle import oracle as O
import bench
k = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sh = S.Shard(0)
tl, th = bench.make_c2_queries(O, 256)
sh.synth_lexical(O.LEX_SEED, int(os.environ.get("DOCS", 10_000_000)), th, O.len_table())
q = sh.make_queries(tl, S.QueryType.Union)
sh.set_strategy(N.BM25_EXHAUSTIVE)
ref = sh.search_lexical_batch(q[:64], k, S.ResultType.Topk, reference_shortcuts=False)
sh.set_strategy(N.BM25_AUTO)
got = sh.search_lexical_batch(q[:64], k, S.ResultType.Topk, reference_shortcuts=False)
assert np.array_equal(ref[1], got[1]) and np.array_equal(ref[2], got[2]), "one-launch answers differ from the exhaustive strategy's"
for nq in (1, 8, 16, 32, 64):
    for rt in (S.ResultType.Topk, S.ResultType.TopkCount):
        for _ in range(3):
            sh.search_lexical_batch(q[:nq], k, rt, reference_shortcuts=False)
        n = 60
        t0 = time.perf_counter()
        for i in range(n):
            o = (i * nq) % (256 - nq + 1)
            sh.search_lexical_batch(q[o:o + nq], k, rt, reference_shortcuts=False)
        print("k %d  %2d queries per call  %-9s %.3f ms per call" % (k, nq, rt.name, (time.perf_counter() - t0) / n * 1e3), flush=True)
print("one-launch batches:", sh.path_stats() if hasattr(sh, "path_stats") else "?")
sh.close()
