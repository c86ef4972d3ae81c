import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import seekstorm_amd as S
from seekstorm_amd import _native as N
from oracle import oracle as O
import test_gpu_pruned as T
n_docs = 150_000
dfs = [0.0004, 0.002, 0.008, 0.02, 0.05, 0.11, 0.3, 0.62, 0.013, 0.004, 0.035, 0.0009]
dl, offs, docs, tfs = T._corpus(O, n_docs, dfs, 5)
sh = S.Shard(0); sh.upload_lexical(n_docs, dl, offs, docs, tfs)
osh = O.Shard(n_docs, dl, offs, docs, tfs)
rng = np.random.default_rng(99)
tl = [[int(x) for x in rng.choice(len(dfs), int(rng.integers(1, 5)), replace=False)] for _ in range(60)]
tl += [[0], [7], [6, 7], [0, 11], [5, 6, 7], [0, 1, 2, 3]]
q = sh.make_queries(tl, S.QueryType.Union)
for k in (1, 10):
    pd, ps, pc, _ = T._run(S, sh, q, k, S.ResultType.Topk, N.BM25_PRUNED)
    ed, es, ec, _ = T._run(S, sh, q, k, S.ResultType.Topk, N.BM25_EXHAUSTIVE)
    bad = [i for i in range(len(tl)) if not (np.array_equal(ps[i], es[i]) and np.array_equal(pd[i], ed[i]))]
    print("k", k, "bad", bad)
    for i in bad[:6]:
        od, os_, _ = osh.search_exhaustive(tl[i], O.OP_OR, k)
        print(" q", i, tl[i], "pruned", pd[i][:3], ps[i][:3].tolist(), "exh", ed[i][:3], es[i][:3].tolist(), "oracle", od[:3], os_[:3].tolist())
        print("   tf max per term", [int(tfs[int(offs[t]):int(offs[t+1])].max()) for t in tl[i]])
