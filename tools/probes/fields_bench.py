"""probe: throughput of queries over SEVERAL indexed fields (title / body / url-like: 3 fields, boosts 2.0 / 1.0 / 0.5) on a
2 M-doc host-built corpus -- 2-term AND top-10 and 3-term OR top-10, 1000 queries per call, TopkCount, AUTO.
Run twice: SS_BM25_MERGED=1 (default: merged per-term lists) and SS_BM25_MERGED=0 (the (term, field) lists only)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import seekstorm_amd as S
from oracle import oracle as O

n_docs, n_fields, n_terms = 2_000_000, 3, 96
rng = np.random.default_rng(7)
lens = np.clip(np.round(np.exp(np.log([12, 300, 8])[:, None] + 0.5 * rng.standard_normal((3, n_docs)))), 1, 60000).astype(np.int64)
lut = np.array([O.lib().so_int_to_byte4(int(x)) for x in range(0, 60001)], np.uint8)
dl = lut[lens]
pf = np.array([0.25, 0.9, 0.15])
dfs = np.concatenate([rng.uniform(0.005, 0.02, 32), rng.uniform(0.02, 0.05, 32), rng.uniform(0.05, 0.15, 32)])
offs, D, F, T = [0], [], [], []
for df in dfs:
    d = np.sort(rng.choice(n_docs, int(df * n_docs), replace=False)).astype(np.uint32)
    m = rng.random((len(d), n_fields)) < pf
    m[~m.any(1), 1] = True
    di, fi = np.nonzero(m)
    D.append(d[di]); F.append(fi.astype(np.uint8)); T.append(np.minimum(rng.geometric(0.5, len(di)), 500).astype(np.uint16))
    offs.append(offs[-1] + len(di))
offs = np.array(offs, np.uint64); D = np.concatenate(D); F = np.concatenate(F); T = np.concatenate(T)
sh = S.Shard(0)
t0 = time.perf_counter()
sh.upload_lexical_fields(n_docs, dl, [2.0, 1.0, 0.5], offs, D, F, T)
print("merged=%s: %d postings, image built in %.1f s" % (os.environ.get("SS_BM25_MERGED", "1"), len(D), time.perf_counter() - t0), flush=True)
lo, mid, hi = np.arange(0, 32), np.arange(32, 64), np.arange(64, 96)
ands = [[int(rng.choice(mid)), int(rng.choice(hi))] for _ in range(1000)]
ors = [[int(rng.choice(lo)), int(rng.choice(mid)), int(rng.choice(hi))] for _ in range(1000)]
for name, tl, qt in (("2-term AND", ands, S.QueryType.Intersection), ("3-term OR", ors, S.QueryType.Union)):
    q = sh.make_queries(tl, qt)
    for rt in (S.ResultType.TopkCount, S.ResultType.Topk):
        sh.search_lexical_batch(q, 10, rt)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 1.0:
            sh.search_lexical_batch(q, 10, rt)
            n += 1
        dt = (time.perf_counter() - t0) / n
        print("  %-10s %-9s %.3f ms / 1000 queries -> %.0f q/s (host pointers, end to end)" % (name, rt.name, dt * 1e3, 1000 / dt), flush=True)
