import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import seekstorm_amd as S
from oracle import oracle as O
import test_gpu_round4 as T
sh, osh, nd, ns, hot, n_docs = T._tiered_shard(S, O)
U = S.QueryType.Union
cases = [([3, nd + 3], [nd + 4]), ([4, 2], [nd + 4]), ([nd + 7, nd + 2, 1], [nd + 3, 0]), ([2], [nd + 4, nd + 7])]
for k in (10, 100):
    print("k", k)
    for terms, nots in cases:
        out = sh.search_lexical_batch(sh.make_queries([terms], U, [nots]), k, reference_shortcuts=False)
        print(" single", terms, nots, int(out[3][0]), osh.search_exhaustive(terms, O.OP_OR, k, not_terms=nots)[2], osh.search_exhaustive(terms, O.OP_OR, k)[2])
    for mixed in ([([0, 1, 2], []), cases[0]], [cases[0], ([0, 1, 2], [])], [cases[0], cases[1]], [([0, 1, 2], []), cases[0], ([nd + 4, 3], []), cases[2], ([4], [2]), cases[3], ([1, nd + 6], [3])]):
        out = sh.search_lexical_batch(sh.make_queries([c[0] for c in mixed], U, [c[1] for c in mixed]), k, reference_shortcuts=False)
        print(" batch", [int(x) for x in out[3]], [osh.search_exhaustive(t, O.OP_OR, k, not_terms=n)[2] for t, n in mixed])
