"""bench.py's `commit` leg: an image with a sparse tier built by commits (commit.rs:142-148 as the seam sees it) -- per 65 536-doc level
the dense terms through ss_bm25_append_level, the rare terms' postings through ss_bm25_append_sparse_level (Shard.commit_level) -- timed
per commit, and its answers compared with a one-shot upload of the same docs (dense image + whole sparse lists).
    run(S, O, thresholds) -> dict;  standalone: python tools/commit_leg.py [levels]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def level_rare_postings(rng, level, n_level_docs, n_sparse, per_level):
    """(list, doc) pairs of one level, low list ids more often than high ones -> CSR over n_sparse lists"""
    key = np.unique(((n_sparse * rng.random(per_level) ** 2).astype(np.uint64) << np.uint64(16)) |
                    rng.integers(0, n_level_docs, per_level).astype(np.uint64))
    lists = (key >> np.uint64(16)).astype(np.int64)
    docs = ((level << 16) + (key & np.uint64(0xFFFF))).astype(np.uint32)
    tfs = np.minimum(rng.geometric(0.6, len(key)), 40).astype(np.uint16)
    offs = np.zeros(n_sparse + 1, np.uint64)
    offs[1:] = np.cumsum(np.bincount(lists, minlength=n_sparse))
    return offs, docs, tfs, lists


def run(S, O, thresholds, n_levels=16, n_dense=256, n_sparse=200_000, per_level=90_000, seed=17, shard_factory=None, last_level_docs=40_000):
    make = shard_factory or (lambda: S.Shard(0))
    n_docs = (n_levels - 1) * 65536 + last_level_docs
    rng = np.random.default_rng(seed)
    dl = O.lex_doclen(n_docs)
    term_ids = np.linspace(0, len(thresholds) - 1, n_dense).astype(np.int64)  # dfs across the generator's whole range (0.05 % .. 20 %)
    dense = [O.lex_term(int(t), thresholds[int(t)], n_docs) for t in term_ids]
    bounds = [np.searchsorted(d, np.arange(n_levels + 1, dtype=np.uint64) * 65536) for d, _ in dense]
    inc = make()
    ms_dense, ms_sparse, rare = [], [], []
    try:
        for lv in range(n_levels):
            lo, hi = lv << 16, min(n_docs, (lv + 1) << 16)
            offs = np.zeros(n_dense + 1, np.uint64)
            offs[1:] = np.cumsum([int(b[lv + 1] - b[lv]) for b in bounds])
            docs = np.concatenate([dense[i][0][bounds[i][lv]:bounds[i][lv + 1]] for i in range(n_dense)]).astype(np.uint32)
            tfs = np.concatenate([dense[i][1][bounds[i][lv]:bounds[i][lv + 1]] for i in range(n_dense)]).astype(np.uint16)
            s_offs, s_docs, s_tfs, s_lists = level_rare_postings(rng, lv, hi - lo, n_sparse, per_level)
            rare.append((s_lists, s_docs, s_tfs))
            t0 = time.perf_counter()
            inc.append_level(lv, dl[lo:hi], offs, docs, tfs)
            t1 = time.perf_counter()
            inc.append_sparse_level(lv, s_offs, s_docs, s_tfs)
            t2 = time.perf_counter()
            ms_dense.append((t1 - t0) * 1e3)
            ms_sparse.append((t2 - t1) * 1e3)
        # the one-shot image of the same docs
        d_offs = np.zeros(n_dense + 1, np.uint64)
        d_offs[1:] = np.cumsum([len(d) for d, _ in dense])
        ref = make()
        try:
            t0 = time.perf_counter()
            ref.upload_lexical(n_docs, dl, d_offs, np.concatenate([d for d, _ in dense]).astype(np.uint32),
                               np.concatenate([t for _, t in dense]).astype(np.uint16))
            lists = np.concatenate([r[0] for r in rare])
            order = np.lexsort((np.concatenate([r[1] for r in rare]), lists))
            r_docs = np.concatenate([r[1] for r in rare])[order]
            r_tfs = np.concatenate([r[2] for r in rare])[order]
            r_offs = np.zeros(n_sparse + 1, np.uint64)
            r_offs[1:] = np.cumsum(np.bincount(lists, minlength=n_sparse))
            ref.append_sparse(r_offs, r_docs, r_tfs)
            one_shot_s = time.perf_counter() - t0
            q_rng = np.random.default_rng(seed + 1)
            queries = [[int(q_rng.integers(0, n_dense)), n_dense + int(q_rng.integers(0, 2000)), int(q_rng.integers(0, n_dense))] for _ in range(24)]
            queries = [list(dict.fromkeys(q)) for q in queries]
            same = True
            for qt in (S.QueryType.Union, S.QueryType.Intersection):
                a = inc.search_lexical_batch(inc.make_queries(queries, qt), 10)
                b = ref.search_lexical_batch(ref.make_queries(queries, qt), 10)
                same = same and all(np.array_equal(x, y) for x, y in zip(a, b))
            tier = inc.sparse_info()
        finally:
            ref.close()
    finally:
        inc.close()
    return {"docs": n_docs, "levels": n_levels, "dense_terms": n_dense, "dense_postings": int(d_offs[-1]), "sparse_lists": n_sparse,
            "sparse_postings": int(tier[1]), "dense_level_ms_median": float(np.median(ms_dense)), "dense_level_ms_last": ms_dense[-1],
            "sparse_level_ms_median": float(np.median(ms_sparse)), "sparse_level_ms_last": ms_sparse[-1],
            "commit_ms_median": float(np.median(np.add(ms_dense, ms_sparse))), "one_shot_upload_s": one_shot_s,
            "answers_equal_one_shot_upload": bool(same), "queries_compared": 2 * len(queries),
            "note": "per commit: ss_bm25_append_level (dense terms: device-side rebuild of the image from the levels kept in HBM) + "
                    "ss_bm25_append_sparse_level (the level's rare postings into their sparse lists, the tier re-coded on the device); "
                    "host clock around the two calls; at the C2 size: profiles/r4y_append_levels_tiered.log"}


if __name__ == "__main__":
    import json
    sys.path.insert(0, ROOT)
    import seekstorm_amd as S
    from oracle import oracle as O
    print(json.dumps(run(S, O, O.term_thresholds(), n_levels=int(sys.argv[1]) if len(sys.argv) > 1 else 16)))
