"""one C2 query at k = 100 / 128 / 256 / 1024 / 4096 through the host-pointer entry: ms per call (which kernels: run under rocprofv3 --kernel-trace --stats)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import seekstorm_amd as S
from oracle import oracle as O
import bench
sh = S.Shard(0)
tl, th = bench.make_c2_queries(O, 16)
sh.synth_lexical(O.LEX_SEED, int(os.environ.get("DOCS", 10_000_000)), th, O.len_table())
q = sh.make_queries(tl, S.QueryType.Union)
ks = [int(x) for x in (sys.argv[1:] or [100, 128, 256, 1024, 4096])]
for k in ks:
    for rt in (S.ResultType.Topk, S.ResultType.TopkCount):
        sh.search_lexical_batch(q[:1], k, rt, reference_shortcuts=False)
        t0 = time.perf_counter()
        n = 30
        for i in range(n):
            sh.search_lexical_batch(q[i % 16:i % 16 + 1], k, rt, reference_shortcuts=False)
        print("k %5d %-9s %.3f ms per query" % (k, rt.name, (time.perf_counter() - t0) / n * 1e3), flush=True)
sh.close()
