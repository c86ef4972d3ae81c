import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import seekstorm_amd as S
from oracle import oracle as O
from test_gpu_parity import _fields_corpus
n_docs, n_fields = 120_000, 3
dfs = [30_000, 9_000, 2_500, 600, 14_000, 0]
dl, offs, docs, fields, tfs = _fields_corpus(O, n_docs, n_fields, dfs, 3 + n_fields)
sh = S.Shard(0)
sh.upload_lexical_fields(n_docs, dl, None, offs, docs, fields, tfs)
print("uploaded", sh.lexical_info(), sh.fields_info(), flush=True)
cases = [([0, 1], []), ([2], []), ([0, 1, 2], []), ([4, 3], [2]), ([1, 4], [0]), ([3], [1]), ([0, 5], []), ([5], [])]
for qt in (S.QueryType.Union, S.QueryType.Intersection):
    for strat in (0, 1):
        sh.set_strategy(strat)
        for rt in (S.ResultType.TopkCount, S.ResultType.Topk, S.ResultType.Count):
            for c in cases:
                print(qt, strat, rt, c, flush=True)
                q = sh.make_queries([c[0]], qt, [c[1]])
                r = sh.search_lexical_batch(q, 10, rt)
                print("  ->", int(r[3][0]), flush=True)
print("batches", flush=True)
for deleted in ((), list(range(5, n_docs, 211))):
    sh.set_deleted(deleted)
    for qt in (S.QueryType.Union, S.QueryType.Intersection):
        for strat in (0, 1):
            sh.set_strategy(strat)
            q = sh.make_queries([c[0] for c in cases], qt, [c[1] for c in cases])
            for rt in (S.ResultType.TopkCount, S.ResultType.Topk, S.ResultType.Count):
                print(bool(deleted), qt, strat, rt, flush=True)
                r = sh.search_lexical_batch(q, 10, rt)
                print("  ->", [int(x) for x in r[3]], flush=True)
