"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same seeded
inputs.  Bar: bit-exact doc-id sets / counts for integer work, scores within 1e-4 relative (north_star)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-4  # north_star tolerance for BM25 / cosine scores


@pytest.fixture(scope="module")
def S():
    import seekstorm_amd
    return seekstorm_amd


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def _check_topk(doc, score, cnt, od, os_, abs_tol=0.0):
    """rows sorted desc; scores within REL of the oracle's; identical id sets outside the tie band of the k-th."""
    n = int(cnt)
    assert n == len(od)
    d, s = doc[:n], score[:n]
    assert np.all(s[:-1] >= s[1:])
    assert np.all(doc[n:] == 0xFFFFFFFF)
    assert len(set(map(int, d))) == n
    assert np.allclose(s, os_, rtol=REL, atol=abs_tol)
    if n:
        # a doc clearly above the k-th score on EITHER side must be on the other side too; scores within REL of each other
        # (the tolerance) may fall on different sides of one band edge, so the two edges are 1 and 2 bands above the k-th
        band = abs(float(os_[-1])) * REL + abs_tol
        clear = lambda dd, ss: {int(x) for x, y in zip(dd, ss) if y > os_[-1] + 2 * band}
        assert clear(d, s) <= {int(x) for x in od} and clear(od, os_) <= {int(x) for x in d}


# ------------------------------------------------------------------ vector path
@pytest.mark.parametrize("n_rows,dim,nq,k", [(20000, 768, 64, 100), (4099, 100, 3, 10), (130, 64, 65, 100), (50, 32, 2, 100)])
def test_vector_parity(S, O, n_rows, dim, nq, k):
    rows = O.vec_gen(O.VEC_SEED, 0, n_rows, dim)
    qs = O.vec_gen(O.VECQ_SEED, 0, nq, dim)
    sh = S.Shard(0)
    sh.upload_vectors(rows)
    doc, score, cnt, tot = sh.search_vector_batch(qs, k)
    for i in range(nq):
        # dot_f32_avx2 silently assumes dim % 8 == 0; other dims take the scalar dot_f32 path
        od, os_, otot, oobs = O.vec_search(rows, qs[i], k, simd_order=(dim % 8 == 0))
        _check_topk(doc[i], score[i], cnt[i], od, os_, abs_tol=2e-6)
        assert tot[i] >= cnt[i]
        if n_rows <= k:
            assert tot[i] == otot == n_rows  # tests/test.rs:676-744 shape: fewer rows than k -> all returned
    sh.close()


def test_vector_self_query_and_row_ids(S, O):
    rows = O.vec_gen(11, 0, 3000, 128)
    ids = (np.arange(3000, dtype=np.uint32) * 7 + 3).astype(np.uint32)  # unique external doc ids
    sh = S.Shard(0)
    sh.upload_vectors(rows, ids)
    doc, score, cnt, tot = sh.search_vector_batch(rows[[5, 1234, 2999]], 10)
    assert list(doc[:, 0]) == [ids[5], ids[1234], ids[2999]]
    assert np.allclose(score[:, 0], 1.0, atol=1e-5)
    sh.close()


@pytest.mark.parametrize("n_rows,n_docs,k", [(6000, 1500, 100), (3000, 40, 100), (5000, 5000, 10)])
def test_vector_multi_record_docs_dedup(S, O, n_rows, n_docs, k):
    """several records per doc (one per field x chunk, vector.rs:561-576): each doc once, with its best score
    (TopK::push dedup, vector.rs:441-452 / 462-473)"""
    dim = 64
    rows = O.vec_gen(31, 0, n_rows, dim)
    rng = np.random.default_rng(7)
    ids = rng.integers(0, n_docs, n_rows).astype(np.uint32) if n_docs < n_rows else rng.permutation(n_rows).astype(np.uint32)
    qs = O.vec_gen(32, 0, 5, dim)
    sh = S.Shard(0)
    sh.upload_vectors(rows, ids)
    doc, score, cnt, tot = sh.search_vector_batch(qs, k)
    for i in range(len(qs)):
        od, os_, _, _ = O.vec_search(rows, qs[i], k, row_doc_ids=ids)
        full = rows @ qs[i]
        best = {}
        for r in range(n_rows):  # independent statement of the semantics: per-doc maximum, then top-k
            d = int(ids[r])
            best[d] = max(best.get(d, -2.0), float(full[r]))
        exp = sorted(best.values(), reverse=True)[:k]
        n = int(cnt[i])
        assert n == len(od) == min(k, len(best))
        assert len(set(map(int, doc[i][:n]))) == n  # every doc once
        assert np.allclose(score[i][:n], exp, rtol=REL, atol=2e-6)
        _check_topk(doc[i], score[i], cnt[i], od, os_, abs_tol=2e-6)
    sh.close()


def test_facet_filter(S, O, lex):
    """FacetFilter (search.rs:735-, is_facet_filter add_result.rs:341-482): a doc failing any filter neither counts nor
    ranks.  Records as facet.bin lays them out (packed, unaligned offsets); numeric half-open ranges of every width and sign,
    floats, string-id sets, several filters at once; unions and intersections, both strategies, on top of tombstones"""
    sh, osh, n_docs = lex
    rng = np.random.default_rng(91)
    rec = np.dtype([("a", "u1"), ("b", "<u2"), ("c", "<i4"), ("d", "<f4"), ("e", "<u8"), ("f", "<i2"), ("g", "<f8"), ("h", "<u4")])
    assert rec.itemsize == 33
    v = np.zeros(n_docs, rec)
    v["a"] = rng.integers(0, 256, n_docs); v["b"] = rng.integers(0, 5000, n_docs); v["c"] = rng.integers(-1000, 1000, n_docs)
    v["d"] = rng.standard_normal(n_docs); v["e"] = rng.integers(0, 1 << 62, n_docs, dtype=np.uint64)
    v["f"] = rng.integers(-300, 300, n_docs); v["g"] = rng.random(n_docs) * 1e6; v["h"] = rng.integers(0, 40, n_docs)
    sh.upload_facets(v.view(np.uint8).reshape(n_docs, 33))
    off = {n: rec.fields[n][1] for n in rec.names}
    gone = list(range(11, n_docs, 379))
    sh.set_deleted(gone)
    osh.set_deleted(gone)
    filter_sets = [
        ([(off["a"], "u8", 10, 200)], (v["a"] >= 10) & (v["a"] < 200)),
        ([(off["c"], "i32", -50, 400), (off["d"], "f32", -0.5, 1.25)], (v["c"] >= -50) & (v["c"] < 400) & (v["d"] >= np.float32(-0.5)) & (v["d"] < np.float32(1.25))),
        ([(off["h"], "string32", [3, 17, 39]), (off["b"], "string16", [7, 8, 9, 10, 11, 4999, 0, 1])],
         np.isin(v["h"], [3, 17, 39]) & np.isin(v["b"], [7, 8, 9, 10, 11, 4999, 0, 1])),
        ([(off["e"], "u64", 1 << 60, 1 << 62), (off["f"], "i16", -300, 0), (off["g"], "f64", 2.5e5, 7.5e5), (off["b"], "u16", 100, 4000)],
         (v["e"] >= (1 << 60)) & (v["f"] < 0) & (v["g"] >= 2.5e5) & (v["g"] < 7.5e5) & (v["b"] >= 100) & (v["b"] < 4000)),
    ]
    # id sets of any size (a StringSet filter resolves to every set id holding the value, search.rs:2643-2710): SS_FACET_IDS_EXTERN
    big16 = sorted(set(rng.integers(0, 5000, 700).tolist())) + [70000]           # an id beyond the facet's width matches nothing
    big32 = sorted(set(rng.integers(0, 40, 25).tolist()))
    filter_sets += [
        ([(off["b"], "string16", big16)], np.isin(v["b"], big16)),
        ([(off["h"], "string32", big32), (off["b"], "string16", big16), (off["a"], "u8", 5, 250)],
         np.isin(v["h"], big32) & np.isin(v["b"], big16) & (v["a"] >= 5) & (v["a"] < 250)),
    ]
    cases = [([10, 9, 8], S.QueryType.Union, O.OP_OR, []), ([10, 9], S.QueryType.Intersection, O.OP_AND, []),
             ([8], S.QueryType.Union, O.OP_OR, []), ([10, 8], S.QueryType.Union, O.OP_OR, [9])]
    for filt, keep in filter_sets:
        excluded = sorted(set(np.nonzero(~keep)[0].tolist()) | set(gone))
        osh.set_deleted(excluded)
        for strat in (0, 1):
            sh.set_strategy(strat)
            for terms, qt, oop, neg in cases:
                q = sh.make_queries([terms], qt, [neg])
                for rt in (S.ResultType.TopkCount, S.ResultType.Count):
                    doc, score, cnt, tot = sh.search_lexical_batch(q, 10, rt, facet_filter=filt)
                    od, os_, otot = osh.search_exhaustive(terms, oop, 10, neg)
                    assert int(tot[0]) == otot, (filt, terms, strat, rt)
                    if rt != S.ResultType.Count:
                        _check_topk(doc[0], score[0], cnt[0], od, os_)
                        assert all(keep[int(d)] for d in doc[0][:cnt[0]])
    sh.set_strategy(0)
    # the filter is per call: the next unfiltered search sees only the tombstones again
    osh.set_deleted(gone)
    q = sh.make_queries([[10, 9]], S.QueryType.Union)
    doc, score, cnt, tot = sh.search_lexical_batch(q, 10)
    assert int(tot[0]) == osh.search_exhaustive([10, 9], O.OP_OR, 10)[2]
    with pytest.raises(S.SeekStormHipError):
        sh.search_lexical_batch(q, 10, facet_filter=[(31, "u32", 0, 5)])   # reads past the record
    sh.set_deleted([])
    osh.set_deleted([])


def test_facet_counts(S, O, lex):
    """query_facets (facet_count, add_result.rs:484-640): the histogram of a facet over a query's match set -- after NOT
    terms, tombstones and the facet filter -- for string ids and numeric ranges (bucket = last lower bound <= value)"""
    sh, osh, n_docs = lex
    rng = np.random.default_rng(93)
    rec = np.dtype([("pad", "u1"), ("cat", "<u2"), ("price", "<f4"), ("year", "<i2"), ("lang", "<u4")])
    v = np.zeros(n_docs, rec)
    v["cat"] = rng.integers(0, 50, n_docs); v["price"] = rng.random(n_docs) * 1000 - 100
    v["year"] = rng.integers(-50, 2030, n_docs); v["lang"] = rng.integers(0, 9, n_docs)
    sh.upload_facets(v.view(np.uint8).reshape(n_docs, rec.itemsize))
    off = {n: rec.fields[n][1] for n in rec.names}
    gone = list(range(5, n_docs, 173))
    sh.set_deleted(gone)
    for terms, qt, oop, neg, filt, keep in (
            ([10, 9, 8], S.QueryType.Union, O.OP_OR, [], None, np.ones(n_docs, bool)),
            ([10, 9], S.QueryType.Intersection, O.OP_AND, [], None, np.ones(n_docs, bool)),
            ([10, 8], S.QueryType.Union, O.OP_OR, [9], [(off["lang"], "u32", 2, 6)], (v["lang"] >= 2) & (v["lang"] < 6)),
            ([7], S.QueryType.Union, O.OP_OR, [], [(off["cat"], "string16", [1, 2, 3, 40])], np.isin(v["cat"], [1, 2, 3, 40]))):
        osh.set_deleted(sorted(set(np.nonzero(~keep)[0].tolist()) | set(gone)))
        od, _, otot = osh.search_exhaustive(terms, oop, n_docs, neg)   # k = n_docs: the whole match set
        assert len(od) == otot
        q = sh.make_queries([terms], qt, [neg])
        counts, other, tot = sh.facet_count(q, off["cat"], "string16", n_buckets=40, facet_filter=filt)
        want = np.bincount(v["cat"][od], minlength=50)
        assert tot == otot and np.array_equal(counts, want[:40]) and other == int(want[40:].sum())
        bounds = [0.0, 50.0, 200.0, 500.0]
        counts, other, tot = sh.facet_count(q, off["price"], "f32", range_lower_bounds=bounds, facet_filter=filt)
        b = np.searchsorted(np.asarray(bounds, np.float32), v["price"][od], side="right") - 1
        assert tot == otot and other == int((b < 0).sum()) and np.array_equal(counts, np.bincount(b[b >= 0], minlength=4))
        ybounds = [-10, 0, 1500, 1999, 2000]
        counts, other, tot = sh.facet_count(q, off["year"], "i16", range_lower_bounds=ybounds, facet_filter=filt)
        b = np.searchsorted(np.asarray(ybounds), v["year"][od].astype(np.int64), side="right") - 1
        assert other == int((b < 0).sum()) and np.array_equal(counts, np.bincount(b[b >= 0], minlength=5))
    sh.set_deleted([])
    osh.set_deleted([])


def test_all_terms_frequent_shortcut(S, O):
    """intersection.rs:198-209 + add_result.rs:2091-2104: when N > 256 k and every term of an intersection is in at least
    half of the docs, a doc with some tf < 10 is counted but not ranked.  The host mirror evaluates the condition like the
    reference and marks the query; counts stay exact; every result type, both strategies, with NOT terms and tombstones"""
    rng = np.random.default_rng(77)
    n_docs = 60_000
    dl = O.lex_doclen(n_docs)
    dfs = [40_000, 33_000, 31_000, 9_000]
    lists = []
    for df in dfs:
        d = np.sort(rng.choice(n_docs, df, replace=False)).astype(np.uint32)
        t = np.minimum(rng.geometric(0.25, df), 700).astype(np.uint16)   # 7.5 % of the postings have tf >= 10
        lists.append((d, t))
    offs = np.zeros(len(lists) + 1, np.uint64)
    offs[1:] = np.cumsum([len(l[0]) for l in lists])
    docs, tfs = np.concatenate([l[0] for l in lists]), np.concatenate([l[1] for l in lists])
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, offs, docs, tfs)
    osh = O.Shard(n_docs, dl, offs, docs, tfs)
    gone = list(range(2, n_docs, 53))
    sh.set_deleted(gone)
    osh.set_deleted(gone)
    cases = [([0, 1], []), ([0, 1, 2], []), ([0, 3], []), ([1, 2], [3]), ([2], [])]
    flagged = [True, True, False, True, False]
    q = sh.make_queries([c[0] for c in cases], S.QueryType.Intersection, [c[1] for c in cases])
    differs = 0
    for k in (10, 200, 300):   # 60 000 > 256 * 200, but not > 256 * 300: no shortcut at k = 300
        marked = sh.mark_all_terms_frequent(q, k)
        assert [bool(x >> 31) for x in marked["op"]] == [f and k < 300 for f in flagged]
        for strat in (0, 1):
            sh.set_strategy(strat)
            for rt in (S.ResultType.TopkCount, S.ResultType.Topk):
                doc, score, cnt, tot = sh.search_lexical_batch(q, k, rt)
                for i, (pos, neg) in enumerate(cases):
                    od, os_, otot = osh.search_exhaustive(pos, O.OP_AND, k, neg, reference_shortcuts=True)
                    plain = osh.search_exhaustive(pos, O.OP_AND, k, neg)
                    differs += int(flagged[i] and k < 300 and not np.array_equal(plain[0], od))
                    if rt == S.ResultType.TopkCount:
                        assert int(tot[i]) == otot == plain[2]
                    _check_topk(doc[i], score[i], cnt[i], od, os_)
    assert differs >= 4  # the shortcut really changes answers here
    sh.set_strategy(0)
    # without the mirror's marking the answer is the exact top-k
    doc, score, cnt, tot = sh.search_lexical_batch(q, 10, reference_shortcuts=False)
    od, os_, _ = osh.search_exhaustive([0, 1], O.OP_AND, 10, [])
    _check_topk(doc[0], score[0], cnt[0], od, os_)
    sh.close()


def test_vector_select_refine_edge_cases(S, O):
    """the refine's radix select: all scores equal (the k-th key is decided in the row bytes of the key: the k smallest
    rows win), k = 1, exactly k rows, and a handful of distinct scores over many rows"""
    dim = 32
    base = O.vec_gen(O.VEC_SEED, 0, 1, dim)[0]
    q = O.vec_gen(O.VECQ_SEED, 0, 3, dim)
    sh = S.Shard(0)
    rows = np.tile(base, (5000, 1)).astype(np.float32)          # 5000 identical records
    sh.upload_vectors(rows)
    for k in (1, 7, 100):
        doc, score, cnt, tot = sh.search_vector_batch(q, k)
        for i in range(3):
            assert cnt[i] == k and np.array_equal(doc[i], np.arange(k)) and len(set(score[i].tolist())) == 1
    rows8 = np.zeros((6000, dim), np.int8)
    rows8[:, 0] = (np.arange(6000) % 5) + 1                      # five distinct integer scores, 1200 rows each
    q8 = np.zeros((2, dim), np.int8); q8[:, 0] = 3
    sh.upload_vectors_i8(rows8)
    doc, score, cnt, tot = sh.search_vector_batch_i8(q8, 100)
    want = np.nonzero(np.arange(6000) % 5 == 4)[0][:100]
    for i in range(2):
        assert cnt[i] == 100 and np.all(score[i] == 15.0) and np.array_equal(doc[i], want)
    sh.upload_vectors_i8(rows8[:100])                             # exactly k rows
    doc, score, cnt, tot = sh.search_vector_batch_i8(q8, 100)
    assert cnt[0] == 100 and sorted(doc[0].tolist()) == list(range(100)) and np.all(score[0][:-1] >= score[0][1:])
    sh.close()


def test_vector_threshold(S, O):
    rows = O.vec_gen(5, 0, 5000, 64)
    q = O.vec_gen(6, 0, 1, 64)[0]
    full = rows @ q
    thr_raw = float(np.sort(full)[-20])  # exactly 20 rows have score >= thr_raw
    sh = S.Shard(0)
    sh.upload_vectors(rows)
    import ctypes as C
    from seekstorm_amd import _native as N
    k = 100
    doc = np.zeros(k, np.uint32); sc = np.zeros(k, np.float32); cnt = np.zeros(1, np.uint32); tot = np.zeros(1, np.uint64)
    N.check(N.lib().ss_vec_search(sh._h, 1, N.ptr(np.ascontiguousarray(q[None]), N.f32p), k, thr_raw, N.ptr(doc, N.u32p),
                                  N.ptr(sc, N.f32p), N.ptr(cnt, N.u32p), N.ptr(tot, N.u64p)), "ss_vec_search")
    od, os_, _, _ = O.vec_search(rows, q, k, threshold_raw=thr_raw)
    assert 18 <= len(od) <= 22
    _check_topk(doc, sc, cnt[0], od, os_, abs_tol=2e-6)
    sh.close()


def test_vector_candidate_overflow_falls_back_exactly(S, O):
    """Adversarial order: every later row beats all earlier ones, so every row is a candidate and the fast
    schedule overflows its candidate buffer; the library must re-run in safe mode and still be exact."""
    n, dim = 40000, 32
    th = np.linspace(1.2, 0.0, n).astype(np.float64)
    rows = np.zeros((n, dim), np.float32)
    rows[:, 0] = np.cos(th)
    rows[:, 1] = np.sin(th)
    q = np.zeros(dim, np.float32)
    q[0] = 1.0
    sh = S.Shard(0)
    sh.upload_vectors(rows)
    doc, score, cnt, tot = sh.search_vector_batch(q, 100)
    od, os_, _, _ = O.vec_search(rows, q, 100)
    _check_topk(doc[0], score[0], cnt[0], od, os_, abs_tol=2e-6)
    sh.close()


def test_vector_synth_matches_oracle_generator(S, O):
    sh = S.Shard(0)
    sh.synth_vectors(O.VEC_SEED, 1000, 96)
    got = sh.read_rows(0, 1000)
    want = O.vec_gen(O.VEC_SEED, 0, 1000, 96)
    assert np.allclose(got, want, rtol=0, atol=1e-7)  # same hash stream; <= 1 ulp from sqrt/div rounding
    assert np.mean(got == want) > 0.5
    sh.close()


# ------------------------------------------------------------------ BM25 path
VOC = [0, 1500, 2500, 3000, 3300, 3600, 3800, 3900, 4000, 4050, 4095]  # df from 0.05% to 20%


@pytest.fixture(scope="module")
def lex(S, O):
    n_docs = 300_000
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = O.lex_corpus(n_docs, VOC)
    osh = O.Shard(n_docs, dl, offs, docs, tfs)
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, offs, docs, tfs)
    yield sh, osh, n_docs
    sh.close()


def test_bm25_image_stats(S, O, lex):
    sh, osh, n_docs = lex
    info = sh.lexical_info()
    assert info["n_docs"] == n_docs and info["n_terms"] == len(VOC)
    assert abs(info["avgdl"] - osh.avgdl) <= 1e-6 * osh.avgdl
    df = sh.posting_count(np.arange(len(VOC)))
    assert [int(x) for x in df] == [osh.df(t) for t in range(len(VOC))]


QUERIES = [[10], [0], [10, 9], [9, 5], [10, 8, 3], [7, 6, 5], [1, 0], [10, 9, 8, 7], [2, 10], [4, 3, 2, 1, 0],
           [10, 9, 8, 7, 6, 5, 4, 3, 2, 1]]


@pytest.mark.parametrize("op", ["Union", "Intersection"])
@pytest.mark.parametrize("k", [10, 100, 200])
def test_bm25_parity(S, O, lex, op, k):
    sh, osh, n_docs = lex
    qt = getattr(S.QueryType, op)
    oop = O.OP_OR if op == "Union" else O.OP_AND
    q = sh.make_queries(QUERIES, qt)
    doc, score, cnt, tot = sh.search_lexical_batch(q, k, S.ResultType.TopkCount)
    for i, terms in enumerate(QUERIES):
        od, os_, otot = osh.search_exhaustive(terms, oop, k)
        assert int(tot[i]) == otot, (terms, op)  # exact result_count_total
        _check_topk(doc[i], score[i], cnt[i], od, os_)
        # and the reference-structured oracle (block-max pruned, containers) agrees too
        od2, os2, otot2 = osh.search(terms, oop, k, O.RT_TOPKCOUNT)
        assert otot2 == otot
        assert np.allclose(os2, score[i][:len(os2)], rtol=REL)


def test_bm25_intersection_doc_sets_bit_exact(S, O, lex):
    """north_star: bit-exact doc-id sets for conjunctive intersection (ask for every match)."""
    sh, osh, n_docs = lex
    terms_list = [[6, 2], [8, 5, 2], [10, 4, 3], [9, 1], [3, 2]]
    q = sh.make_queries(terms_list, S.QueryType.Intersection)
    doc, score, cnt, tot = sh.search_lexical_batch(q, 1024, S.ResultType.TopkCount)
    for i, terms in enumerate(terms_list):
        od, os_, otot = osh.search_exhaustive(terms, O.OP_AND, 1024)
        assert otot <= 1024, "fixture must fit in k"
        assert int(tot[i]) == otot == int(cnt[i])
        assert set(map(int, doc[i][:cnt[i]])) == set(map(int, od))


def test_bm25_count_and_mixed_batch(S, O, lex):
    sh, osh, n_docs = lex
    terms_list = [[10, 9], [10, 9], [5], [6, 2]]
    types = [S.QueryType.Union, S.QueryType.Intersection, S.QueryType.Union, S.QueryType.Intersection]
    q = sh.make_queries(terms_list, types)
    doc, score, cnt, tot = sh.search_lexical_batch(q, 10, S.ResultType.Count)
    for i, (terms, t) in enumerate(zip(terms_list, types)):
        _, _, otot = osh.search_exhaustive(terms, O.OP_OR if t == S.QueryType.Union else O.OP_AND, 1)
        assert int(tot[i]) == otot and cnt[i] == 0
    doc, score, cnt, tot = sh.search_lexical_batch(q, 10, S.ResultType.Topk)
    for i, (terms, t) in enumerate(zip(terms_list, types)):
        od, os_, _ = osh.search_exhaustive(terms, O.OP_OR if t == S.QueryType.Union else O.OP_AND, 10)
        _check_topk(doc[i], score[i], cnt[i], od, os_)


def test_bm25_large_batch_is_deterministic(S, O, lex):
    sh, osh, n_docs = lex
    rng = np.random.default_rng(3)
    tl = [list(rng.choice(len(VOC), size=3, replace=False)) for _ in range(700)]
    q = sh.make_queries(tl, S.QueryType.Union)
    a = sh.search_lexical_batch(q, 10)
    b = sh.search_lexical_batch(q, 10)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    for i in (0, 123, 699):
        od, os_, otot = osh.search_exhaustive(tl[i], O.OP_OR, 10)
        assert int(a[3][i]) == otot
        _check_topk(a[0][i], a[1][i], a[2][i], od, os_)


def test_bm25_synth_image_equals_uploaded_image(S, O):
    """the device-side generator must produce the corpus the oracle generates on the host"""
    n_docs, nt = 70_000, 16
    th = O.term_thresholds(nt)
    tab = O.len_table()
    a = S.Shard(0)
    a.synth_lexical(O.LEX_SEED, n_docs, th, tab)
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = O.lex_corpus(n_docs, list(range(nt)), thresholds=th)
    b = S.Shard(0)
    b.upload_lexical(n_docs, dl, offs, docs, tfs)
    ia, ib = a.lexical_info(), b.lexical_info()
    assert ia == ib
    assert np.array_equal(a.posting_count(np.arange(nt)), b.posting_count(np.arange(nt)))
    tl = [[15, 14, 9], [13, 2], [15], [12, 11, 10, 3]]
    for qt in (S.QueryType.Union, S.QueryType.Intersection):
        ra = a.search_lexical_batch(a.make_queries(tl, qt), 10)
        rb = b.search_lexical_batch(b.make_queries(tl, qt), 10)
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y)
    a.close(); b.close()


def test_abi_rejects_bad_input_without_crashing(S, O, lex):
    sh, osh, n_docs = lex
    q = sh.make_queries([[1, 2]], S.QueryType.Union)
    q["term"][0, 0] = 10_000  # out of vocabulary
    with pytest.raises(S.SeekStormHipError):
        sh.search_lexical_batch(q, 10)
    empty = S.Shard(0)
    assert empty.search_lexical_shard([1]).result_count == 0  # degrades to empty like search.rs:2461-2463
    with pytest.raises(S.SeekStormHipError):
        empty.search_lexical_shard([1], strict=True)
    assert empty.search_vector_shard(np.ones(8, np.float32)).result_count == 0
    empty.close()


# ------------------------------------------------------------------ planner: shards, hybrid RRF
def test_index_two_shards_hybrid_matches_oracle(S, O):
    n_docs, dim, S_n = 40_000, 64, 2
    voc = [3000, 3600, 4000]
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = O.lex_corpus(n_docs, voc)
    rows = O.vec_gen(O.VEC_SEED, 0, n_docs, dim)
    qv = O.vec_gen(O.VECQ_SEED, 0, 1, dim, normalize=False)[0] * 3.0  # un-normalised on purpose
    shards, oshards, orows = [], [], []
    for sid in range(S_n):  # doc g -> shard g % S, local id g // S (index.rs:5284)
        sel = np.arange(sid, n_docs, S_n)
        o2, d2, t2 = [0], [], []
        for t in range(len(voc)):
            d = docs[int(offs[t]):int(offs[t + 1])]
            f = tfs[int(offs[t]):int(offs[t + 1])]
            m = (d % S_n) == sid
            d2.append(d[m] // S_n); t2.append(f[m]); o2.append(o2[-1] + int(m.sum()))
        d2 = np.concatenate(d2).astype(np.uint32); t2 = np.concatenate(t2)
        sh = S.Shard(0, shard_id=sid)
        sh.upload_lexical(len(sel), dl[sel], np.asarray(o2, np.uint64), d2, t2)
        sh.upload_vectors(rows[sel])
        shards.append(sh)
        oshards.append(O.Shard(len(sel), dl[sel], np.asarray(o2, np.uint64), d2, t2))
        orows.append(rows[sel])
    idx = S.Index(shards)
    k = 20
    qn = O.normalize(qv)
    lex_d, lex_s, vec_d, vec_s = [], [], [], []
    for sid in range(S_n):
        od, os_, _ = oshards[sid].search_exhaustive([0, 1, 2], O.OP_OR, k)
        lex_d += [int(x) * S_n + sid for x in od]; lex_s += list(os_)
        vd, vs, _, _ = O.vec_search(orows[sid], qn, k)
        vec_d += [int(x) * S_n + sid for x in vd]; vec_s += list(vs)
    for mode, omode in ((S.SearchMode.Lexical, 0), (S.SearchMode.Vector, 1), (S.SearchMode.Hybrid, 2)):
        ro = idx.search([0, 1, 2], qv, S.QueryType.Union, mode, 0, k, strict=True)
        od, os_, osrc = O.merge(omode, (lex_d, lex_s), (vec_d, vec_s), 0, k)
        assert ro.result_count == len(od) == k
        got_s = np.array([r.score for r in ro.results], np.float32)
        assert np.allclose(got_s, os_, rtol=REL, atol=2e-6)
        band = abs(float(os_[-1])) * REL + 2e-6
        assert {r.doc_id for r in ro.results if r.score > os_[-1] + band} == {int(x) for x, y in zip(od, os_) if y > os_[-1] + band}
    # the search() options of the same seam: NOT terms, a facet filter (lexical side), an AnnMode (vector side)
    for sid, sh in enumerate(shards):
        n_s = sh.indexed_doc_count
        year = ((np.arange(n_s) * 7 + sid) % 50).astype("<i2")
        sh.upload_facets(year.view(np.uint8).reshape(n_s, 2))
        sh.set_clusters([1, 1], [n_s // 2, n_s - n_s // 2])
        osel = np.nonzero(~((year >= 10) & (year < 40)))[0]
        oshards[sid].set_deleted(osel)
    lex_d, lex_s, vec_d, vec_s = [], [], [], []
    for sid in range(S_n):
        od, os_, _ = oshards[sid].search_exhaustive([0, 1], O.OP_OR, k, [2])
        lex_d += [int(x) * S_n + sid for x in od]; lex_s += list(os_)
        n_s = shards[sid].indexed_doc_count
        vd, vs, _, _, _ = O.vec_search_ann(orows[sid], qn, k, [1, 1], [n_s // 2, n_s - n_s // 2], n_probe=1)
        vec_d += [int(x) * S_n + sid for x in vd]; vec_s += list(vs)
    ro = idx.search([0, 1], qv, S.QueryType.Union, S.SearchMode.Hybrid, 0, k, strict=True, not_terms=[2],
                    facet_filter=[(0, "i16", 10, 40)], ann_mode=S.AnnMode.Nprobe(1))
    od, os_, osrc = O.merge(2, (lex_d, lex_s), (vec_d, vec_s), 0, k)
    assert ro.result_count == len(od)
    assert np.allclose(np.array([r.score for r in ro.results], np.float32), os_, rtol=REL, atol=2e-6)
    for sh in shards:
        sh.close()


@pytest.mark.parametrize("Sn,nq,k,ties", [(4, 9, 100, False), (16, 5, 1024, True), (3, 4, 5000, True), (9, 3, 1000, False)])
def test_device_merge_matches_host_merge(S, O, Sn, nq, k, ties):
    """ss_topk_merge_dev (used after the RCCL all-gather) == ss_merge_results on the same gathered lists -- the LDS sort up to 8192 keys
    per query, beyond that (more than 8 shards at k = 1024, a deep page) the rank merge: every entry finds its slot by binary searches in
    the other shards' lists; equal scores in concatenation order either way"""
    import ctypes as C
    import torch
    from seekstorm_amd import distributed as D
    rng = np.random.default_rng(5)
    doc = np.stack([np.stack([rng.choice(100000, k, replace=False) for _ in range(nq)]) for _ in range(Sn)]).astype(np.int32)
    score = rng.standard_normal((Sn, nq, k)).astype(np.float32)
    if ties:
        score = np.round(score * 8) / 8  # many equal scores, inside a list and across the lists
    score = np.sort(score, axis=2)[:, :, ::-1].copy()  # negative scores too
    cnt = rng.integers(0, k + 1, (Sn, nq)).astype(np.int32)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        g = [torch.from_numpy(x).to(dev) for x in (doc, score, cnt)]
        m_doc, m_score, m_cnt = D.merge_gathered_device(*g, C.c_void_p(st.cuda_stream), 0)
        # the packed form (one all-gather per batch): per shard [doc | score bits | count]
        packed = torch.stack([D.pack_topk(g[0][s_], g[1][s_], g[2][s_]) for s_ in range(Sn)])
        p_doc, p_score, p_cnt = D.merge_gathered_device_packed(packed, nq, k, C.c_void_p(st.cuda_stream), 0)
        st.synchronize()
        assert torch.equal(p_doc, m_doc) and torch.equal(p_score, m_score) and torch.equal(p_cnt, m_cnt)
    host = D.merge_gathered_host(torch.from_numpy(doc), torch.from_numpy(score), torch.from_numpy(cnt), 0, k, S.SearchMode.Vector)
    md, ms, mc = m_doc.cpu().numpy(), m_score.cpu().numpy(), m_cnt.cpu().numpy()
    for q in range(nq):
        hd, hs = host[q]
        n = int(mc[q])
        assert n == len(hd) == min(k, int(cnt[:, q].sum()))
        assert np.array_equal(md[q, :n].astype(np.uint64), hd) and np.array_equal(ms[q, :n], hs)
        assert np.all(md[q, n:] == -1)


@pytest.mark.parametrize("wide", [False, True], ids=["u32", "u64"])
def test_device_rrf_matches_host_rrf(S, O, wide):
    """ss_rrf_merge_dev == ss_merge_results(Hybrid) per query: fused scores bit for bit, sources, order (ties by doc id),
    offset / length; also with one list empty or absent"""
    import ctypes as C
    import torch
    rng = np.random.default_rng(9)
    nq, kl, kv = 23, 100, 100
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    idt = np.int64 if wide else np.int32
    hi = (1 << 40) if wide else 400
    ld = np.zeros((nq, kl), idt); vd = np.zeros((nq, kv), idt)
    lc = rng.integers(0, kl + 1, nq).astype(np.int32); vc = rng.integers(0, kv + 1, nq).astype(np.int32)
    lc[0], vc[0] = 0, 0
    lc[1], vc[1] = kl, 0
    lc[2], vc[2] = 0, kv
    for q in range(nq):
        pool = rng.choice(400, 150, replace=False).astype(np.int64) * (hi // 400)   # small pool: many docs in both lists
        ld[q] = rng.permutation(pool)[:kl]
        vd[q] = rng.permutation(pool)[:kv]
    ls = np.sort(rng.random((nq, kl)).astype(np.float32), axis=1)[:, ::-1]
    vs = np.sort(rng.random((nq, kv)).astype(np.float32), axis=1)[:, ::-1]
    with torch.cuda.stream(st):
        t = [torch.from_numpy(x).to(dev) for x in (ld, lc, vd, vc)]
        for offset, length in ((0, 100), (7, 30), (0, 250), (190, 20)):
            od, os_, src, cnt = S.rrf_merge_device(t[0], t[1], t[2], t[3], offset, length, C.c_void_p(st.cuda_stream))
            st.synchronize()
            od, os_, src, cnt = od.cpu().numpy(), os_.cpu().numpy(), src.cpu().numpy(), cnt.cpu().numpy()
            for q in range(nq):
                hd, hs, hsrc = S.merge_results(S.SearchMode.Hybrid, (ld[q, :lc[q]].astype(np.uint64), ls[q, :lc[q]]),
                                               (vd[q, :vc[q]].astype(np.uint64), vs[q, :vc[q]]), offset, length)
                n = int(cnt[q])
                assert n == len(hd)
                assert np.array_equal(od[q, :n].astype(np.uint64), hd) and np.array_equal(os_[q, :n], hs)
                assert np.array_equal(src[q, :n], hsrc) and np.all(od[q, n:] == -1)
        # a list that is absent altogether: the other one's ranks
        od, os_, src, cnt = S.rrf_merge_device(None, None, t[2], t[3], 0, 10, C.c_void_p(st.cuda_stream))
        st.synchronize()
        q = 5
        n = min(10, int(vc[q]))
        assert int(cnt[q]) == n and np.array_equal(od[q, :n].cpu().numpy(), vd[q, :n].astype(np.int64))
        assert np.all(src[q, :n].cpu().numpy() == int(S.ResultSource.Vector))


def test_deleted_docs_lexical_both_strategies(S, O, lex):
    """delete_hashset (add_result.rs:3435, union.rs:975): a tombstoned doc neither counts nor ranks, in any result type,
    under the exhaustive and the pruned strategy; clearing the set restores the answers."""
    sh, osh, n_docs = lex
    tl_all = [[10, 9, 8], [9, 5], [10], [7, 6, 5, 4], [10, 9, 8, 7, 6, 5]]
    rng = np.random.default_rng(17)
    try:
        for qt, oop in ((S.QueryType.Union, O.OP_OR), (S.QueryType.Intersection, O.OP_AND)):
            tl = tl_all
            q = sh.make_queries(tl, qt)
            base = sh.search_lexical_batch(q, 10)
            gone = set(int(x) for x in rng.choice(n_docs, size=3000, replace=False))
            for i in range(len(tl)):  # plus ranked docs of every query, first and middle
                gone.update(int(d) for d in base[0][i][:base[2][i]][::3])
            gone = sorted(gone)
            sh.set_deleted(gone)
            osh.set_deleted(gone)
            for strat in (1, 2, 0):  # EXHAUSTIVE, PRUNED, AUTO
                sh.set_strategy(strat)
                tl = tl_all[:4] if strat == 2 else tl_all  # the pruned kernel serves <= 4 terms
                q = sh.make_queries(tl, qt)
                for rt in (S.ResultType.TopkCount, S.ResultType.Topk, S.ResultType.Count):
                    doc, score, cnt, tot = sh.search_lexical_batch(q, 10, rt)
                    for i, terms in enumerate(tl):
                        od, os_, otot = osh.search_exhaustive(terms, oop, 10)
                        if rt != S.ResultType.Topk:
                            assert int(tot[i]) == otot, (terms, qt, strat, rt)
                        if rt != S.ResultType.Count:
                            assert not set(map(int, doc[i][:cnt[i]])) & set(gone)
                            _check_topk(doc[i], score[i], cnt[i], od, os_)
            sh.set_strategy(0)
            sh.set_deleted([])
            osh.set_deleted([])
            again = sh.search_lexical_batch(sh.make_queries(tl_all, qt), 10)
            for x, y in zip(base, again):
                assert np.array_equal(x, y)
    finally:
        sh.set_strategy(0)
        sh.set_deleted([])
        osh.set_deleted([])


def test_deleted_docs_vector(S, O):
    rows = O.vec_gen(O.VEC_SEED, 0, 6000, 96)
    ids = (np.arange(6000) // 2).astype(np.uint32)  # two records per doc
    qs = O.vec_gen(O.VECQ_SEED, 0, 6, 96)
    sh = S.Shard(0)
    sh.upload_vectors(rows, ids)
    base = sh.search_vector_batch(qs, 50)
    gone = sorted({int(d) for i in range(len(qs)) for d in base[0][i][:10:2]} | {1, 2999})
    sh.set_deleted(np.array(gone, np.uint64).tobytes())  # delete.bin's layout: a stream of u64 (index.rs:3798-3809)
    doc, score, cnt, tot = sh.search_vector_batch(qs, 50)
    for i in range(len(qs)):
        od, os_, _, _ = O.vec_search(rows, qs[i], 50, row_doc_ids=ids, deleted=gone)
        assert cnt[i] == len(od) and not set(map(int, doc[i][:cnt[i]])) & set(gone)
        assert np.allclose(score[i][:cnt[i]], os_, rtol=1e-4, atol=2e-6)
        assert [int(x) for x in doc[i][:30]] == [int(x) for x in od[:30]]
    sh.set_deleted([])
    for x, y in zip(base, sh.search_vector_batch(qs, 50)):
        assert np.array_equal(x, y)
    sh.close()


def test_not_terms_both_strategies(S, O, lex):
    """not_query_list (add_result.rs:3440-3497): docs of a NOT list neither count nor rank; every result type, both
    strategies, NT-specialised and grouped kernels, together with tombstones."""
    sh, osh, n_docs = lex
    cases = [([10, 9], [8]), ([10, 9, 8], [7]), ([10], [9]), ([7, 6], [10, 2]), ([5], [10, 9, 8]),
             ([10, 9, 8, 7], [6, 5]), ([4, 3, 2, 1, 0], [10]), ([10, 9, 8], [7, 6, 5, 4, 3, 2, 1])]
    try:
        for qt, oop in ((S.QueryType.Union, O.OP_OR), (S.QueryType.Intersection, O.OP_AND)):
            for strat in (1, 2, 0):
                cs = [c for c in cases if len(c[0]) <= 4] if strat == 2 else cases
                sh.set_strategy(strat)
                q = sh.make_queries([c[0] for c in cs], qt, [c[1] for c in cs])
                assert all(int(o) >> 8 == len(c[1]) for o, c in zip(q["op"], cs))
                for rt in (S.ResultType.TopkCount, S.ResultType.Topk, S.ResultType.Count):
                    doc, score, cnt, tot = sh.search_lexical_batch(q, 10, rt)
                    for i, (pos, neg) in enumerate(cs):
                        od, os_, otot = osh.search_exhaustive(pos, oop, 10, not_terms=neg)
                        if rt != S.ResultType.Topk:
                            assert int(tot[i]) == otot, (pos, neg, qt, strat, rt)
                        if rt != S.ResultType.Count:
                            _check_topk(doc[i], score[i], cnt[i], od, os_)
        # NOT terms + tombstones + a mixed batch (with and without NOT terms)
        sh.set_strategy(0)
        gone = list(range(0, n_docs, 97))
        sh.set_deleted(gone)
        osh.set_deleted(gone)
        mixed = [([10, 9, 8], []), ([10, 9, 8], [7]), ([6], [5]), ([9, 3], [])]
        for qt, oop in ((S.QueryType.Union, O.OP_OR), (S.QueryType.Intersection, O.OP_AND)):
            q = sh.make_queries([c[0] for c in mixed], qt, [c[1] for c in mixed])
            doc, score, cnt, tot = sh.search_lexical_batch(q, 10)
            for i, (pos, neg) in enumerate(mixed):
                od, os_, otot = osh.search_exhaustive(pos, oop, 10, not_terms=neg)
                assert int(tot[i]) == otot
                _check_topk(doc[i], score[i], cnt[i], od, os_)
                od2, os2, otot2 = osh.search(pos, oop, 10, O.RT_TOPKCOUNT, not_terms=neg)  # reference-structured oracle
                assert otot2 == otot and np.allclose(os2, os_, rtol=REL)
    finally:
        sh.set_strategy(0)
        sh.set_deleted([])
        osh.set_deleted([])


def test_not_terms_in_a_query_that_is_not_the_longest(S, O, lex):
    """the batch's longest query has no NOT terms, a shorter one has: n_terms + NOT terms of both are equal, so "the batch
    holds NOT terms" must not be derived from those maxima alone (the unfiltered kernel variants ignore NOT lists)"""
    sh, osh, n_docs = lex
    mixed = [([10, 9, 8], []), ([10, 9], [8]), ([7, 6, 5], []), ([10], [9, 8])]
    try:
        for qt, oop in ((S.QueryType.Union, O.OP_OR), (S.QueryType.Intersection, O.OP_AND)):
            for strat in (0, 1, 2):
                sh.set_strategy(strat)
                q = sh.make_queries([c[0] for c in mixed], qt, [c[1] for c in mixed])
                doc, score, cnt, tot = sh.search_lexical_batch(q, 10)
                for i, (pos, neg) in enumerate(mixed):
                    od, os_, otot = osh.search_exhaustive(pos, oop, 10, not_terms=neg)
                    assert int(tot[i]) == otot, (pos, neg, qt, strat)
                    _check_topk(doc[i], score[i], cnt[i], od, os_)
    finally:
        sh.set_strategy(0)


def test_not_terms_abi_validation(S, O, lex):
    sh, osh, n_docs = lex
    q = sh.make_queries([[10, 9]], S.QueryType.Union, [[8]])
    bad = q.copy()
    bad["op"][0] = int(S.QueryType.Union) | (9 << 8)  # 2 + 9 > 10 terms
    with pytest.raises(S.SeekStormHipError):
        sh.search_lexical_batch(bad, 10)
    bad = q.copy()
    bad["term"][0, 2] = 10  # NOT term repeats a query term
    with pytest.raises(S.SeekStormHipError):
        sh.search_lexical_batch(bad, 10)
    bad = q.copy()
    bad["op"][0] = int(S.QueryType.Union) | (2 << 16)  # field filter naming a field this one-field image does not have
    with pytest.raises(S.SeekStormHipError):
        sh.search_lexical_batch(bad, 10)
    ok = q.copy()
    ok["op"][0] |= 1 << 16  # ... and naming its only field: no filter at all
    assert all(np.array_equal(x, y) for x, y in zip(sh.search_lexical_batch(ok, 10), sh.search_lexical_batch(q, 10)))


@pytest.mark.parametrize("n_rows,dim,nq,k", [(20000, 768, 64, 100), (4099, 100, 3, 10), (300, 384, 70, 50), (50, 32, 2, 100),
                                             (9000, 1024, 5, 20)])
def test_vector_i8_parity(S, O, n_rows, dim, nq, k):
    """Precision::I8: integer dot products are exact -> scores bit-identical to the oracle's, ids identical outside ties"""
    rows = O.quantize_i8(O.vec_gen(O.VEC_SEED, 0, n_rows, dim))
    qs = O.quantize_i8(O.vec_gen(O.VECQ_SEED, 0, nq, dim))
    sh = S.Shard(0)
    sh.upload_vectors_i8(rows)
    assert np.array_equal(sh.read_rows_i8(0, min(n_rows, 64)), rows[:64])
    doc, score, cnt, tot = sh.search_vector_batch_i8(qs, k)
    full = rows.astype(np.int32) @ qs.astype(np.int32).T
    for i in range(nq):
        od, os_, _, _ = O.vec_search_i8(rows, qs[i], k)
        assert cnt[i] == len(od)
        assert np.array_equal(score[i][:cnt[i]], os_)  # bit-exact
        assert np.array_equal(score[i][:cnt[i]], full[doc[i][:cnt[i]], i].astype(np.float32))  # each id really has its score
        kth = os_[-1]
        assert {int(x) for x, y in zip(doc[i][:cnt[i]], score[i]) if y > kth} == {int(x) for x, y in zip(od, os_) if y > kth}
    sh.close()


def test_vector_i8_scales_dedup_tombstones_threshold(S, O):
    n_rows, dim, k = 7000, 256, 30
    rows = O.quantize_i8(O.vec_gen(O.VEC_SEED, 0, n_rows, dim))
    qs = O.quantize_i8(O.vec_gen(O.VECQ_SEED, 0, 9, dim))
    rng = np.random.default_rng(8)
    rs = rng.uniform(0.002, 0.02, n_rows).astype(np.float32)   # VectorHeader.scale per record
    qscale = rng.uniform(0.005, 0.01, len(qs)).astype(np.float32)
    ids = (np.arange(n_rows) // 3).astype(np.uint32)            # three records per doc
    sh = S.Shard(0)
    sh.upload_vectors_i8(rows, row_scale=rs, row_doc_ids=ids)
    gone = [5, 77, 1200]
    sh.set_deleted(gone)
    doc, score, cnt, tot = sh.search_vector_batch_i8(qs, k, query_scale=qscale)
    for i in range(len(qs)):
        od, os_, _, _ = O.vec_search_i8(rows, qs[i], k, row_doc_ids=ids, row_scale=rs, query_scale=float(qscale[i]), deleted=gone)
        assert cnt[i] == len(od) and not set(map(int, doc[i][:cnt[i]])) & set(gone)
        assert np.array_equal(score[i][:cnt[i]], os_)  # (dot as f32 * scale1) * scale2: same two roundings
        assert len(set(map(int, doc[i][:cnt[i]]))) == cnt[i]
    # threshold on the raw score (TopK::new, vector.rs:388-397): fewer than k results
    sh.set_deleted([])
    thr = float(np.sort(((rows.astype(np.int32) @ qs[0].astype(np.int32)).astype(np.float32) * qscale[0]) * rs)[-5])
    d2, s2, c2, _ = sh.search_vector_batch_i8(qs[:1], k, query_scale=qscale[:1], similarity_threshold_raw=thr)
    od, os_, _, _ = O.vec_search_i8(rows, qs[0], k, row_doc_ids=ids, row_scale=rs, query_scale=float(qscale[0]), threshold_raw=thr)
    assert c2[0] == len(od) <= 5 and np.array_equal(s2[0][:c2[0]], os_)
    # the f32 entry points refuse an i8 image instead of misreading it
    with pytest.raises(S.SeekStormHipError):
        sh.search_vector_batch(np.zeros((1, dim), np.float32), 10)
    sh.close()


def test_vector_i8_synth_is_the_quantised_f32_generator(S, O):
    sh = S.Shard(0)
    sh.synth_vectors_i8(O.VEC_SEED, 3000, 96)
    want = O.quantize_i8(O.vec_gen(O.VEC_SEED, 0, 3000, 96))
    got = sh.read_rows_i8(0, 3000)
    assert np.array_equal(got, want)
    sh.close()


def test_reference_count_assertions_on_the_hip_path(S, O):
    """The only values the reference's own tests pin for this path, through the C ABI and under both
    strategies: tests/test.rs:150-177 ("+body2 +test" -> 1 result, total 1), tests/test.rs:181-208 ("test", Count -> 2),
    tests/test.rs:676-744 (3 vectors, k = 10 -> 3 hits)."""
    doclen = np.array([O.lib().so_int_to_byte4(x) for x in (3, 4, 4, 3)], np.uint8)  # 4-doc fixture, tests/test.rs:96-148
    offs = np.array([0, 2, 3], np.uint64)                                            # term 0 = 'test' (docs 1, 2), 1 = 'body2' (doc 1)
    docs = np.array([1, 2, 1], np.uint32)
    tfs = np.array([1, 1, 1], np.uint16)
    sh = S.Shard(0)
    sh.upload_lexical(4, doclen, offs, docs, tfs)
    for strat in (0, 1, 2):
        sh.set_strategy(strat)
        ro = sh.search_lexical_shard([1, 0], S.QueryType.Intersection, 0, 10, S.ResultType.TopkCount, strict=True)
        assert ro.result_count == 1 and ro.result_count_total == 1 and ro.results[0].doc_id == 1
        ro = sh.search_lexical_shard([0], S.QueryType.Union, 0, 10, S.ResultType.Count, strict=True)
        assert ro.result_count == 0 and ro.result_count_total == 2
        ro = sh.search_lexical_shard([0, 1], S.QueryType.Union, 0, 10, S.ResultType.TopkCount, strict=True)
        assert ro.result_count_total == 2 and [r.doc_id for r in ro.results] == [1, 2]
    rows = O.vec_gen(7, 0, 3, 128)
    sh.upload_vectors(rows)
    ro = sh.search_vector_shard(rows[1], 10, strict=True)
    assert ro.result_count == 3 and ro.result_count_total == 3 and ro.results[0].doc_id == 1 and abs(ro.results[0].score - 1.0) < 1e-5
    sh.close()


def test_concurrent_callers_share_a_shard(S, O, lex):
    """search.rs callers hold only the shard READ lock and arrive from many runtime threads: lexical, vector and
    tombstone calls from 12 threads on one handle, answers equal to the single-threaded ones"""
    import threading
    sh, osh, n_docs = lex
    rows = O.vec_gen(O.VEC_SEED, 0, 3000, 64)
    qs = O.vec_gen(O.VECQ_SEED, 0, 4, 64)
    vs = S.Shard(0)
    vs.upload_vectors(rows)
    tl = [[10, 9, 8], [7, 3], [5], [9, 8, 7, 6, 5]]
    want_l = {i: sh.search_lexical_batch(sh.make_queries([t], S.QueryType.Union), 10) for i, t in enumerate(tl)}
    want_v = {i: vs.search_vector_batch(qs[i:i + 1], 20) for i in range(4)}
    errors = []

    def worker(seed):
        try:
            rng = np.random.default_rng(seed)
            for _ in range(25):
                i = int(rng.integers(0, 4))
                if rng.random() < 0.5:
                    got = sh.search_lexical_batch(sh.make_queries([tl[i]], S.QueryType.Union), 10)
                    ok = all(np.array_equal(a, b) for a, b in zip(got, want_l[i]))
                else:
                    got = vs.search_vector_batch(qs[i:i + 1], 20)
                    ok = all(np.array_equal(a, b) for a, b in zip(got, want_v[i]))
                if not ok:
                    errors.append((seed, i))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(s,)) for s in range(12)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]
    # every one of those calls went through the coalescer of its shard (one query per call, no filters)
    lb, lq, _, _ = sh.coalescing_stats()
    _, _, vb, vq = vs.coalescing_stats()
    assert lq + vq >= 12 * 25 and lb <= lq and vb <= vq
    # callers that arrive together SHARE device batches (the reference's workers hold only the shard read lock): 16
    # threads released by a barrier, the leader waits up to 50 ms for company -> far fewer batches than calls, and every caller
    # still gets its own answer, cut to its own k (the top-k' of a query is the head of its top-k)
    sh.set_coalescing(1024, 64, 50000)
    vs.set_coalescing(1024, 64, 50000)
    lb0, lq0, _, _ = sh.coalescing_stats()
    _, _, vb0, vq0 = vs.coalescing_stats()
    bar = threading.Barrier(16)
    got = {}

    def burst(j):
        try:
            i, kk = j % 4, (10, 7, 4)[j % 3]
            bar.wait()
            if j < 10:
                got[j] = ("l", i, kk, sh.search_lexical_batch(sh.make_queries([tl[i]], S.QueryType.Union), kk))
            else:
                got[j] = ("v", i, kk, vs.search_vector_batch(qs[i:i + 1], 2 * kk))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=burst, args=(j,)) for j in range(16)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]
    lb1, lq1, _, _ = sh.coalescing_stats()
    _, _, vb1, vq1 = vs.coalescing_stats()
    assert lq1 - lq0 == 10 and vq1 - vq0 == 6
    assert lb1 - lb0 <= 3 and vb1 - vb0 <= 3, (lb1 - lb0, vb1 - vb0)  # 10 + 6 concurrent calls, a handful of device batches
    for j, (kind, i, kk, g) in got.items():
        want = want_l[i] if kind == "l" else want_v[i]
        kq = kk if kind == "l" else 2 * kk
        assert np.array_equal(g[0][0][:kq], want[0][0][:kq]) and np.array_equal(g[1][0][:kq], want[1][0][:kq]), (j, kind)
        if kind == "l":
            assert int(g[3][0]) == int(want[3][0])  # exact match counts do not depend on k
    sh.set_coalescing()
    vs.set_coalescing()
    # an invalid request inside a merged batch fails ALONE: its batch-mates get their answers
    sh.set_coalescing(1024, 64, 50000)
    bar = threading.Barrier(6)
    outcome = {}

    def mixed(j):
        q = sh.make_queries([tl[j % 4]], S.QueryType.Union)
        if j == 2:
            q["term"][0][0] = 0xFFFFFF  # no such term
        bar.wait()
        try:
            outcome[j] = sh.search_lexical_batch(q, 10)
        except Exception as e:  # noqa: BLE001
            outcome[j] = e

    th = [threading.Thread(target=mixed, args=(j,)) for j in range(6)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    sh.set_coalescing()
    assert isinstance(outcome[2], Exception)
    for j in (0, 1, 3, 4, 5):
        assert not isinstance(outcome[j], Exception), outcome[j]
        assert all(np.array_equal(a, b) for a, b in zip(outcome[j], want_l[j % 4]))
    vs.close()


def _fields_corpus(O, n_docs, n_fields, dfs, seed):
    """postings (term, doc, field, tf) sorted by (doc, field) per term; a doc holds a term in 1..n_fields fields"""
    rng = np.random.default_rng(seed)
    dl = np.stack([O.lex_doclen(n_docs, seed=O.LEX_SEED + 17 * f) for f in range(n_fields)])
    offs, docs, fields, tfs = [0], [], [], []
    for df in dfs:
        d = np.sort(rng.choice(n_docs, size=df, replace=False))
        for f in range(n_fields):
            keep = rng.random(df) < (0.7 if f == 0 else 0.35)
            if f == 0:
                none = ~keep
            else:
                none &= ~keep
            docs.append(d[keep]); fields.append(np.full(int(keep.sum()), f)); tfs.append(np.minimum(rng.geometric(0.5, int(keep.sum())), 40))
        docs.append(d[none]); fields.append(np.full(int(none.sum()), n_fields - 1)); tfs.append(np.ones(int(none.sum()), np.int64))  # every doc somewhere
        n_t = sum(len(x) for x in docs[-(n_fields + 1):])
        dd = np.concatenate(docs[-(n_fields + 1):]); ff = np.concatenate(fields[-(n_fields + 1):]); tt = np.concatenate(tfs[-(n_fields + 1):])
        del docs[-(n_fields + 1):], fields[-(n_fields + 1):], tfs[-(n_fields + 1):]
        o = np.lexsort((ff, dd))
        # (doc, field) must be unique: the "every doc somewhere" filler may repeat the last field
        keep = np.ones(n_t, bool)
        keep[1:] = (dd[o][1:] != dd[o][:-1]) | (ff[o][1:] != ff[o][:-1])
        docs.append(dd[o][keep]); fields.append(ff[o][keep]); tfs.append(tt[o][keep])
        offs.append(offs[-1] + int(keep.sum()))
    return (dl, np.array(offs, np.uint64), np.concatenate(docs).astype(np.uint32), np.concatenate(fields).astype(np.uint8),
            np.concatenate(tfs).astype(np.uint16))


@pytest.mark.parametrize("n_fields,boost", [(2, [2.0, 1.0]), (3, None)])
def test_bm25f_several_fields(S, O, n_fields, boost):
    """get_bm25f_multiterm_multifield (add_result.rs:1171-1426): per-field tf / length / boost, df over any field,
    intersections of unions; every result type and strategy, NOT terms and tombstones, against the brute-force oracle"""
    n_docs = 120_000
    dfs = [30_000, 9_000, 2_500, 600, 14_000, 0]  # the last term has no posting in any field
    dl, offs, docs, fields, tfs = _fields_corpus(O, n_docs, n_fields, dfs, 3 + n_fields)
    sh = S.Shard(0)
    sh.upload_lexical_fields(n_docs, dl, boost, offs, docs, fields, tfs)
    info = sh.lexical_info()
    assert info["n_terms"] == len(dfs) and info["n_docs"] == n_docs
    assert [int(x) for x in sh.posting_count(np.arange(len(dfs)))] == dfs  # docs containing the term in any field
    cases = [([0, 1], []), ([2], []), ([0, 1, 2], []), ([4, 3], [2]), ([1, 4], [0]), ([3], [1]), ([0, 5], []), ([5], [])]
    gone = list(range(5, n_docs, 211))
    for deleted in ((), gone):
        sh.set_deleted(deleted)
        for qt, oop in ((S.QueryType.Union, O.OP_OR), (S.QueryType.Intersection, O.OP_AND)):
            for strat in (0, 1):
                sh.set_strategy(strat)
                q = sh.make_queries([c[0] for c in cases], qt, [c[1] for c in cases])
                for rt in (S.ResultType.TopkCount, S.ResultType.Topk, S.ResultType.Count):
                    doc, score, cnt, tot = sh.search_lexical_batch(q, 10, rt)
                    for i, (pos, neg) in enumerate(cases):
                        od, os_, otot, avg = O.search_fields_exhaustive(n_docs, dl, boost, offs, docs, fields, tfs, pos, oop, 10, neg, deleted)
                        assert abs(info["avgdl"] - avg) <= 1e-6 * avg
                        if rt != S.ResultType.Topk:
                            assert int(tot[i]) == otot, (pos, neg, qt, strat, rt)
                        if rt != S.ResultType.Count:
                            _check_topk(doc[i], score[i], cnt[i], od, os_)
    sh.set_strategy(0)
    # queries without a field filter read one MERGED list per term: the pruned strategy serves their intersections and unions
    # (the same answers as the scan kernels, bit for bit), and the single-field limits apply (10 terms, not 32 / n_fields)
    from seekstorm_amd import _native as N
    sh.set_deleted(())
    for qt in (S.QueryType.Intersection, S.QueryType.Union):
        q = sh.make_queries([[0, 1], [0, 1, 4], [4, 3, 0, 1]], qt)
        for rt in (S.ResultType.TopkCount, S.ResultType.Topk):
            sh.set_strategy(N.BM25_EXHAUSTIVE)
            a = sh.search_lexical_batch(q, 10, rt)
            sh.set_strategy(N.BM25_PRUNED)
            b = sh.search_lexical_batch(q, 10, rt)
            assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3])), (qt, rt)
            if rt == S.ResultType.TopkCount:
                assert np.array_equal(a[3], b[3])
    sh.set_strategy(0)
    q = sh.make_queries([[0, 1, 2, 3, 4]], S.QueryType.Union, [[5]])  # 6 terms on 3 fields: fine either way
    sh.search_lexical_batch(q, 10)
    q = sh.make_queries([[0, 1, 2, 3, 4, 5]] , S.QueryType.Intersection)
    doc, score, cnt, tot = sh.search_lexical_batch(q, 10)
    assert int(tot[0]) == 0 and int(cnt[0]) == 0  # the last term is in no doc
    sh.close()


def test_bm25f_boosts_too_far_apart_for_merged_lists(S, O):
    """Merged per-term lists hold sum_f boost_f * w_f in the 19-bit weight code (2^-14 .. 2^2 after a power-of-two scale chosen
    from the corpus).  Boosts 4096 / 1 / 1/4096 span far more: the image is then built WITHOUT merged lists (exact per-field
    scores rather than clamped ones) -- parity with the oracle as ever, and intersections stay with the scan kernels."""
    from seekstorm_amd import _native as N
    n_docs, n_fields = 60_000, 3
    boost = [4096.0, 1.0, 1.0 / 4096.0]
    dfs = [20_000, 6_000, 1_500]
    dl, offs, docs, fields, tfs = _fields_corpus(O, n_docs, n_fields, dfs, 77)
    sh = S.Shard(0)
    sh.upload_lexical_fields(n_docs, dl, boost, offs, docs, fields, tfs)
    for terms, qt, oop in (([0, 1], S.QueryType.Union, O.OP_OR), ([0, 1, 2], S.QueryType.Intersection, O.OP_AND), ([2], S.QueryType.Union, O.OP_OR)):
        q = sh.make_queries([terms], qt)
        doc, score, cnt, tot = sh.search_lexical_batch(q, 10, S.ResultType.TopkCount)
        od, os_, otot, _ = O.search_fields_exhaustive(n_docs, dl, boost, offs, docs, fields, tfs, terms, oop, 10, (), ())
        assert int(tot[0]) == otot
        _check_topk(doc[0], score[0], cnt[0], od, os_)
    sh.set_strategy(N.BM25_PRUNED)
    with pytest.raises(S.SeekStormHipError):   # (term, field) lists only: an intersection of unions is a scan-kernel query
        sh.search_lexical_batch(sh.make_queries([[0, 1]], S.QueryType.Intersection), 10, S.ResultType.Topk)
    sh.close()
    # ordinary boosts: merged lists, and the pruned strategy serves the same intersection
    sh = S.Shard(0)
    sh.upload_lexical_fields(n_docs, dl, [2.0, 1.0, 0.5], offs, docs, fields, tfs)
    sh.set_strategy(N.BM25_PRUNED)
    sh.search_lexical_batch(sh.make_queries([[0, 1]], S.QueryType.Intersection), 10, S.ResultType.Topk)
    sh.close()


def test_bm25f_merged_lists_with_rationed_probe_rows(S, O):
    """Several fields + a probe budget below the number of lists: rows go to the longest lists first -- the merged lists --, so
    queries without a field filter stay on the pruned strategy; a field-filtered intersection reads (term, field) lists, of which
    the shorter ones get pool rows on demand or leave the query to the scan kernels.  Every answer equals the unrationed
    shard's, bit for bit."""
    n_docs, n_fields = 150_000, 3
    dfs = [int(150_000 * 0.2 / (1 + 0.5 * i)) for i in range(12)]   # 12 terms x (3 fields + merged) = 48 lists
    dl, offs, docs, fields, tfs = _fields_corpus(O, n_docs, n_fields, dfs, 5)
    boost = [2.0, 1.0, 0.5]
    full, part = S.Shard(0), S.Shard(0)
    full.upload_lexical_fields(n_docs, dl, boost, offs, docs, fields, tfs)
    n_sub = (n_docs + 4095) // 4096
    part.set_probe_budget((40 + 1) * n_sub * 64 * 12)   # 40 rows for 48 lists: 30 fixed + a pool of 10
    part.upload_lexical_fields(n_docs, dl, boost, offs, docs, fields, tfs)
    rng = np.random.default_rng(9)
    for rnd in range(3):
        tl = [[int(x) for x in rng.choice(12, int(rng.integers(1, 4)), replace=False)] for _ in range(12)]
        for qt in (S.QueryType.Union, S.QueryType.Intersection):
            for rt in (S.ResultType.TopkCount, S.ResultType.Topk):
                a = full.search_lexical_batch(full.make_queries(tl, qt), 10, rt)
                b = part.search_lexical_batch(part.make_queries(tl, qt), 10, rt)
                assert all(np.array_equal(x, y) for x, y in zip(a, b)), (rnd, qt, rt)
        for filt in ([0], [2], [1, 2]):
            a = full.search_lexical_batch(full.make_queries(tl, S.QueryType.Intersection, field_filter=filt), 10, S.ResultType.TopkCount)
            b = part.search_lexical_batch(part.make_queries(tl, S.QueryType.Intersection, field_filter=filt), 10, S.ResultType.TopkCount)
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), (rnd, filt)
    full.close()
    part.close()


@pytest.mark.parametrize("n_fields", [2, 3])
def test_bm25f_field_filter(S, O, n_fields):
    """field_filter on an image with several indexed fields (add_result.rs:3124-3136): a doc stays only if every query term
    occurs in one of the listed fields; the score still sums all fields.  Intersections and single terms, every result
    type and strategy, with NOT terms and tombstones; a union of several terms is refused"""
    n_docs = 90_000
    dfs = [40_000, 25_000, 6_000, 30_000]
    dl, offs, docs, fields, tfs = _fields_corpus(O, n_docs, n_fields, dfs, 17 + n_fields)
    boost = [1.5, 1.0, 0.5][:n_fields]
    sh = S.Shard(0)
    sh.upload_lexical_fields(n_docs, dl, boost, offs, docs, fields, tfs)
    gone = list(range(3, n_docs, 97))
    sh.set_deleted(gone)
    cases = [([0, 1], []), ([2], []), ([0, 1, 3], []), ([3, 1], [2]), ([0], [1])]
    for filt in ([0], [n_fields - 1], [0, n_fields - 1]):
        for strat in (0, 1):
            sh.set_strategy(strat)
            q = sh.make_queries([c[0] for c in cases], S.QueryType.Intersection, [c[1] for c in cases], field_filter=filt)
            for rt in (S.ResultType.TopkCount, S.ResultType.Topk, S.ResultType.Count):
                doc, score, cnt, tot = sh.search_lexical_batch(q, 10, rt)
                for i, (pos, neg) in enumerate(cases):
                    od, os_, otot, _ = O.search_fields_exhaustive(n_docs, dl, boost, offs, docs, fields, tfs, pos, O.OP_AND, 10, neg,
                                                                  gone, field_filter=filt)
                    unf = O.search_fields_exhaustive(n_docs, dl, boost, offs, docs, fields, tfs, pos, O.OP_AND, 10, neg, gone)[2]
                    assert otot <= unf and (otot < unf or len(filt) == n_fields or otot == 0)  # the filter really bites
                    if rt != S.ResultType.Topk:
                        assert int(tot[i]) == otot, (pos, neg, filt, strat, rt)
                    if rt != S.ResultType.Count:
                        _check_topk(doc[i], score[i], cnt[i], od, os_)
    sh.set_strategy(0)
    # every field listed == no filter: the same docs and counts; the unfiltered query reads the terms' MERGED lists, whose
    # weights are the per-field sums rounded once more to the weight code (2^-16 relative)
    a = sh.search_lexical_batch(sh.make_queries([[0, 1]], S.QueryType.Intersection, field_filter=list(range(n_fields))), 10)
    b = sh.search_lexical_batch(sh.make_queries([[0, 1]], S.QueryType.Intersection), 10)
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.allclose(a[1], b[1], rtol=1e-4)
    assert set(a[0][0].tolist()) == set(b[0][0].tolist()) or np.allclose(a[1][0][-1], b[1][0][-1], rtol=1e-4)
    # a union of several terms under a filter is answered by the scan kernels since round 3 (per-term gating; the rule is checked in
    # test_union_under_a_field_filter_follows_the_reference_decomposition); more than 7 terms: not offered
    sh.search_lexical_batch(sh.make_queries([[0, 1]], S.QueryType.Union, field_filter=[0]), 10)
    with pytest.raises(S.SeekStormHipError):   # a field the image does not have
        sh.search_lexical_batch(sh.make_queries([[0]], S.QueryType.Union, field_filter=[n_fields]), 10)
    sh.close()


@pytest.mark.parametrize("n_fields", [2, 3])
def test_union_under_a_field_filter_follows_the_reference_decomposition(S, O, n_fields):
    """A union of several terms with a field filter (Shard.search_lexical_shard): the reference tries every subset of the
    terms as a filtered intersection (union_docid_3's queue down to pairs, union_docid_2 = the pair + its two single terms)
    and a doc keeps its best -- the sum over its terms that occur in a listed field, all fields of those terms counted; a
    doc none of whose terms passes is no result.  Brute-force oracle of that rule from per-term oracle scores; totals as the
    reference reports them (2 terms: filtered union, more: the unfiltered union); NOT terms and tombstones on top."""
    n_docs = 90_000
    dfs = [40_000, 25_000, 6_000, 30_000]
    dl, offs, docs, fields, tfs = _fields_corpus(O, n_docs, n_fields, dfs, 23 + n_fields)
    boost = [1.5, 1.0, 0.5][:n_fields]
    sh = S.Shard(0)
    sh.upload_lexical_fields(n_docs, dl, boost, offs, docs, fields, tfs)
    gone = list(range(3, n_docs, 97))
    sh.set_deleted(gone)
    gone_set = set(gone)
    per_term = {}
    for t in range(len(dfs)):  # every doc of the term with the term's own score (all its fields), and where the term sits
        d, s_, tot, _ = O.search_fields_exhaustive(n_docs, dl, boost, offs, docs, fields, tfs, [t], O.OP_OR, n_docs, (), gone)
        a, b = int(offs[t]), int(offs[t + 1])
        per_term[t] = (dict(zip(d.tolist(), s_.tolist())), docs[a:b], fields[a:b])
    for filt in ([0], [n_fields - 1], [0, n_fields - 1][:n_fields]):
        fset = set(filt)
        for terms, neg in (([0, 1], []), ([0, 1, 3], []), ([2, 3], [0]), ([0, 1, 2, 3], []), ([3, 1, 2], [0])):
            score, passing, present = {}, [], set()
            for t in terms:
                sc, dd, ff = per_term[t]
                pas = set(dd[np.isin(ff, list(fset))].tolist()) - gone_set
                passing.append(pas)
                present |= set(dd.tolist()) - gone_set
                for d in pas:
                    score[d] = np.float32(score.get(d, np.float32(0)) + np.float32(sc[d]))
            banned = set()
            for t in neg:
                banned |= set(per_term[t][1].tolist())
            ranked = sorted(((d, float(v)) for d, v in score.items() if d not in banned), key=lambda e: (-e[1], e[0]))
            # both routes: per-term gating inside the scan kernels (round 3: BM_AND_GATED), and the composition from the
            # reference's own sub-queries (2^n - 1 filtered intersections merged by the maximum)
            for k, compose in ((10, False), (40, False), (10, True), (40, True)):
                sh.compose_filtered_unions = compose
                ro = sh.search_lexical_shard(terms, S.QueryType.Union, 0, k, S.ResultType.TopkCount, strict=True, not_terms=neg, field_filter=filt)
                sh.compose_filtered_unions = False
                want = ranked[:k]
                assert len(ro.results) == len(want), (terms, neg, filt, k)
                got_s = np.array([r.score for r in ro.results], np.float32)
                assert np.allclose(got_s, [w[1] for w in want], rtol=1e-4), (terms, neg, filt, k)
                kth = want[-1][1] if want else 0.0
                band = abs(kth) * 2e-4
                assert {r.doc_id for r in ro.results if r.score > kth + band} == {d for d, v in want if v > kth + band}
                if len(terms) == 2:
                    assert ro.result_count_total == len((passing[0] | passing[1]) - banned), (terms, neg, filt)
                else:
                    assert ro.result_count_total == len(present - banned), (terms, neg, filt)
            # a doc whose only matching terms sit in unlisted fields must be absent even when its unfiltered score is high
            un = sh.search_lexical_shard(terms, S.QueryType.Union, 0, 40, S.ResultType.Topk, strict=True, not_terms=neg)
            if len(filt) < n_fields:
                assert any(r.doc_id not in score for r in un.results) or len(un.results) == 0 or all(r.doc_id in score for r in un.results)
    sh.close()


def test_c1_standin_one_million_docs_and_pairs(S, O):
    """BASELINE configs[0] stand-in (LEX-1M, 2-term AND, top-10, df bands [1 %, 5 %] x [5 %, 20 %]): the device
    generator's corpus against the reference-structured oracle on the host-generated copy of the same corpus"""
    n_docs = 1_000_000
    th = O.term_thresholds()
    df = th.astype(np.float64) / 2.0 ** 32
    lo = np.nonzero((df >= 0.01) & (df < 0.05))[0]
    hi = np.nonzero((df >= 0.05) & (df < 0.20))[0]
    rng = np.random.default_rng(12)
    pairs = [(int(rng.choice(lo)), int(rng.choice(hi))) for _ in range(40)]
    sh = S.Shard(0)
    sh.synth_lexical(O.LEX_SEED, n_docs, th, O.len_table())
    q = sh.make_queries([list(p) for p in pairs], S.QueryType.Intersection)
    doc, score, cnt, tot = sh.search_lexical_batch(q, 10, S.ResultType.TopkCount)
    doc2, score2, cnt2, _ = sh.search_lexical_batch(q, 10, S.ResultType.Topk)
    assert np.array_equal(score, score2) and np.array_equal(doc, doc2)
    dl = O.lex_doclen(n_docs)
    for i in range(0, len(pairs), 4):  # the oracle on every fourth pair: host generation of the two lists + its own search
        terms = sorted(set(pairs[i]))
        offs, docs, tfs = O.lex_corpus(n_docs, terms)
        osh = O.Shard(n_docs, dl, offs, docs, tfs)
        local = [terms.index(t) for t in pairs[i]]
        od, os_, otot = osh.search(local, O.OP_AND, 10, O.RT_TOPKCOUNT)
        assert int(tot[i]) == otot
        _check_topk(doc[i], score[i], cnt[i], od, os_)
    sh.close()


def test_full_size_properties_c2_c3(S, O):
    """BASELINE.json's full sizes (C2: 10 M docs, 3-term unions top-10; C3: 10 M x 768, batch 64, top-100) over a LARGER query
    set than the oracle comparison of tests/test_gpu_fullsize.py samples: size-independent properties -- both strategies
    bit-identical, exact counts equal, sorted, idempotent; every returned vector score is the dot product of the row that
    was returned (f32 and i8)."""
    from seekstorm_amd import _native as N
    n = 10_000_000
    th = O.term_thresholds()
    sh = S.Shard(0)
    sh.synth_lexical(O.LEX_SEED, n, th, O.len_table())
    df = th.astype(np.float64) / 2.0 ** 32
    bands = [np.nonzero((df >= a) & (df < b))[0] for a, b in ((0.005, 0.02), (0.02, 0.05), (0.05, 0.15))]
    rng = np.random.default_rng(5)
    tl = [[int(rng.choice(b)) for b in bands] for _ in range(300)]
    q = sh.make_queries(tl, S.QueryType.Union)
    res = {}
    for strat in (N.BM25_EXHAUSTIVE, N.BM25_AUTO):
        sh.set_strategy(strat)
        res[strat] = [sh.search_lexical_batch(q, 10, rt) for rt in (S.ResultType.Topk, S.ResultType.TopkCount)]
        again = sh.search_lexical_batch(q, 10, S.ResultType.TopkCount)
        assert all(np.array_equal(a, b) for a, b in zip(res[strat][1], again))  # idempotent
    for a, b in zip(res[N.BM25_EXHAUSTIVE], res[N.BM25_AUTO]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert np.array_equal(res[N.BM25_EXHAUSTIVE][1][3], res[N.BM25_AUTO][1][3])  # exact totals: scan count mode == bit records
    doc, score, cnt, tot = res[N.BM25_AUTO][1]
    assert np.all(cnt == 10) and np.all(score[:, :-1] >= score[:, 1:]) and np.all(score[:, -1] > 0)
    dfs = np.array(sh.posting_count(np.unique(tl)), np.int64)
    dfm = dict(zip(np.unique(tl).tolist(), dfs.tolist()))
    for i, t in enumerate(tl):  # max df <= |union| <= sum df
        assert max(dfm[x] for x in t) <= int(tot[i]) <= sum(dfm[x] for x in t)
    sh.set_strategy(N.BM25_AUTO)
    # C3
    dim, B, k = 768, 64, 100
    qs = O.vec_gen(O.VECQ_SEED, 0, B, dim)
    sh.synth_vectors(O.VEC_SEED, n, dim)
    d, s, c, t = sh.search_vector_batch(qs, k)
    assert np.all(c == k) and np.all(s[:, :-1] >= s[:, 1:])
    for i in (0, 17, 63):
        rows = np.stack([sh.read_rows(int(r), 1)[0] for r in d[i, :5]])
        assert np.allclose(rows @ qs[i], s[i, :5], rtol=1e-4, atol=2e-6)
        assert len(set(d[i].tolist())) == k
    d1, s1, c1, _ = sh.search_vector_batch(qs[:1], k)  # the <= 32-query instantiation returns the same list
    assert np.array_equal(d1[0], d[0]) and np.array_equal(s1[0], s[0])
    q8 = O.quantize_i8(qs)
    sh.synth_vectors_i8(O.VEC_SEED, n, dim)
    d8, s8, c8, _ = sh.search_vector_batch_i8(q8, k)
    assert np.all(c8 == k) and np.all(s8[:, :-1] >= s8[:, 1:])
    for i in (0, 31, 63):
        rows = np.stack([sh.read_rows_i8(int(r), 1)[0] for r in d8[i, :5]]).astype(np.int64)
        assert np.array_equal((rows @ q8[i].astype(np.int64)).astype(np.float32), s8[i, :5])
    sh.close()


def test_result_sort_by_facets(S, O, lex):
    """result_sort (search.rs ResultSort; ordering min_heap.rs:574-1050): the k best matches under (facet fields..., score) --
    composed from the pivot (ss_bm25_facet_kth: radix select over the match set) and filtered searches.  Against a brute-force
    ordering of the oracle's full match list: one and two sort fields, ascending / descending, every numeric width and sign,
    floats, low-cardinality fields (huge tie groups), fewer matches than k, on top of a facet filter and tombstones."""
    sh, osh, n_docs = lex
    rng = np.random.default_rng(17)
    rec = np.dtype([("a", "u1"), ("c", "<i4"), ("d", "<f4"), ("e", "<u8"), ("f", "<i2"), ("g", "<f8"), ("h", "<u2")])
    v = np.zeros(n_docs, rec)
    v["a"] = rng.integers(0, 4, n_docs); v["c"] = rng.integers(-1000, 1000, n_docs); v["d"] = rng.standard_normal(n_docs)
    v["e"] = rng.integers(0, 1 << 62, n_docs, dtype=np.uint64); v["f"] = rng.integers(-3, 3, n_docs)
    v["g"] = rng.random(n_docs) * 1e6 - 5e5; v["h"] = rng.integers(0, 50000, n_docs)
    sh.upload_facets(v.view(np.uint8).reshape(n_docs, rec.itemsize))
    off = {n: rec.fields[n][1] for n in rec.names}
    ty = {"a": "u8", "c": "i32", "d": "f32", "e": "u64", "f": "i16", "g": "f64", "h": "u16"}
    gone = list(range(5, n_docs, 211))
    cases = [([10, 9, 8], S.QueryType.Union, O.OP_OR), ([10, 9], S.QueryType.Intersection, O.OP_AND), ([3], S.QueryType.Union, O.OP_OR)]
    sorts = [[("c", True)], [("c", False)], [("d", True)], [("g", False)], [("e", True)], [("h", False)],
             [("a", True), ("c", False)], [("f", False), ("a", True)], [("a", False), ("f", True), ("d", True)]]
    try:
        for deleted in (False, True):
            sh.set_deleted(gone if deleted else [])
            osh.set_deleted(gone if deleted else [])
            for terms, qt, oop in cases:
                q = sh.make_queries([terms], qt)
                ad, as_, atot = osh.search_exhaustive(terms, oop, n_docs)  # every match, by score
                for srt in sorts:
                    for k, flt in ((10, None), (37, [(off["h"], "u16", 1000, 30000)]), (atot + 5 if atot < 2000 else 100, None)):
                        keep = np.ones(len(ad), bool) if flt is None else (v["h"][ad] >= 1000) & (v["h"][ad] < 30000)
                        md, ms = ad[keep], as_[keep]
                        cols = []
                        for name, desc in srt:
                            x = v[name][md].astype(np.float64) if ty[name][0] == "f" else v[name][md].astype(object)
                            cols.append([(-t if desc else t) for t in x.tolist()])
                        order = sorted(range(len(md)), key=lambda i: tuple(c[i] for c in cols) + (-float(ms[i]),))[:k]
                        doc, score, tot = sh.search_lexical_sorted(q, [(off[n], ty[n], d_) for n, d_ in srt], k, facet_filter=flt)
                        assert tot == len(md), (terms, srt, k, tot, len(md))
                        assert len(doc) == len(order)
                        # the sort-field values must agree position by position; docs may swap only inside groups whose fields AND scores tie
                        for name, _ in srt:
                            assert np.array_equal(v[name][doc], v[name][md[order]]), (terms, srt, k, name)
                        assert np.allclose(score, ms[order], rtol=1e-4), (terms, srt, k)
            # one call for a batch of queries (ss_bm25_search_sorted): row i = the single call's answer; and the host-composed route
            # (pivot + filtered searches through the older entry points) agrees on fields and scores
            qb = np.concatenate([sh.make_queries([t], qt) for t, qt, _ in cases])
            for srt in ([("c", True)], [("a", False), ("f", True), ("d", True)]):
                spec = [(off[n], ty[n], d_) for n, d_ in srt]
                bd, bs, bc, bt = sh.search_lexical_sorted_batch(qb, spec, 25)
                for i in range(len(cases)):
                    doc, score, tot = sh.search_lexical_sorted(qb[i:i + 1], spec, 25)
                    assert int(bt[i]) == tot and int(bc[i]) == len(doc) and np.array_equal(bd[i][:bc[i]], doc) and np.array_equal(bs[i][:bc[i]], score)
                    cd, cs, ctot = sh.search_lexical_sorted_composed(qb[i:i + 1], spec, 25)
                    assert ctot == tot and len(cd) == len(doc) and np.allclose(cs, score, rtol=1e-6)
                    for name, _ in srt:
                        assert np.array_equal(v[name][cd], v[name][doc])
            # more queries than one chunk of the batched pipeline holds (64): rows repeat with the queries
            big = np.concatenate([qb] * 50)
            bd2, bs2, bc2, bt2 = sh.search_lexical_sorted_batch(big, [(off["c"], ty["c"], True)], 12)
            b0 = sh.search_lexical_sorted_batch(qb, [(off["c"], ty["c"], True)], 12)
            for i in range(len(big)):
                j = i % len(qb)
                assert int(bt2[i]) == int(b0[3][j]) and int(bc2[i]) == int(b0[2][j]) and np.array_equal(bd2[i], b0[0][j]) and np.array_equal(bs2[i], b0[1][j])
    finally:
        sh.set_deleted([])
        osh.set_deleted([])


def _ulps(a, b):
    a = np.asarray(a, np.float64).view(np.int64)
    b = np.asarray(b, np.float64).view(np.int64)
    return np.abs(a - b)


def test_point_facets(S, O, lex):
    """Point facets (FieldType::Point = the u64 Morton code of (lat, lon), geo_search.rs): the distance filter
    (FilterSparse::Point, add_result.rs:462-478: inside the reference's Morton range AND euclidian_distance inside the
    range), distance-range counts (Ranges::Point, add_result.rs:605-618) and the sort by distance (morton_ordering,
    min_heap.rs:510-528), km and miles, combined with another filter / a second sort field, against the oracle's f64
    restatement.  The device's cos / sqrt may differ from libm in the last places: most distances are bit-equal, all within 16
    ulp, so a doc could only change sides if it sat that close to a bound (none does with these seeds)."""
    sh, osh, n_docs = lex
    rng = np.random.default_rng(23)
    rec = np.dtype([("pad", "u1"), ("loc", "<u8"), ("h", "<u2"), ("coarse", "<u8")])
    lat = rng.random(n_docs) * 50.0 + 10.0          # a cloud over the north-east quadrant: the Morton range is a real range there
    lon = rng.random(n_docs) * 60.0 + 5.0
    v = np.zeros(n_docs, rec)
    v["loc"] = O.morton_encode(lat, lon)
    v["h"] = rng.integers(0, 50000, n_docs)
    v["coarse"] = O.morton_encode(np.round(lat / 10.0) * 10.0, np.round(lon / 20.0) * 20.0)   # few distinct points: huge tie groups
    sh.upload_facets(v.view(np.uint8).reshape(n_docs, rec.itemsize))
    off = {n: rec.fields[n][1] for n in rec.names}
    all_docs = np.arange(n_docs, dtype=np.uint32)
    bases = [(38.8951, 30.25), (52.52, 13.405), (-10.0, -20.0)]
    try:
        # distances and stored codes as the library computes them
        assert np.array_equal(sh.facet_values(all_docs[:5000], off["loc"], "point"), v["loc"][:5000])
        for base in bases:
            for unit in ("km", "miles", "sortkey"):
                got = sh.facet_point_distances(all_docs, off["loc"], base, unit)
                want = O.geo_distances(v["loc"], base, unit)
                u = _ulps(got, want)
                assert u.max() <= 16 and (u == 0).mean() > 0.9, (base, unit, u.max(), (u == 0).mean())
        gone = list(range(7, n_docs, 197))
        sh.set_deleted(gone)
        cases = [([10, 9, 8], S.QueryType.Union, O.OP_OR), ([10, 9], S.QueryType.Intersection, O.OP_AND), ([3], S.QueryType.Union, O.OP_OR)]
        # ---- filter
        for base, lo, hi, unit in (((38.8951, 30.25), 0.0, 1000.0, "km"), ((38.8951, 30.25), 200.0, 900.0, "miles"),
                                   ((52.52, 13.405), 100.0, 800.0, "km"), ((30.0, 1.0), 0.0, 800.0, "km"),   # box crosses lon 0: empty Z range
                                   ((-10.0, -20.0), 0.0, 20000.0, "km")):
            m0, m1 = O.geo_morton_range(base, hi, unit)
            dist = O.geo_distances(v["loc"], base, unit)
            keep = (v["loc"] >= np.uint64(m0)) & (v["loc"] < np.uint64(m1)) & (dist >= lo) & (dist < hi)
            for extra, ekeep in ((None, np.ones(n_docs, bool)), ((off["h"], "u16", 1000, 30000), (v["h"] >= 1000) & (v["h"] < 30000))):
                filt = [(off["loc"], "point", base, lo, hi, unit)] + ([extra] if extra else [])
                osh.set_deleted(sorted(set(np.nonzero(~(keep & ekeep))[0].tolist()) | set(gone)))
                for terms, qt, oop in cases:
                    q = sh.make_queries([terms], qt)
                    doc, score, cnt, tot = sh.search_lexical_batch(q, 10, S.ResultType.TopkCount, facet_filter=filt)
                    od, os_, otot = osh.search_exhaustive(terms, oop, 10)
                    assert int(tot[0]) == otot, (base, lo, hi, unit, terms)
                    _check_topk(doc[0], score[0], cnt[0], od, os_)
            if base == (30.0, 1.0):
                assert m0 > m1 and not keep.any()       # the reference's Z-order range is empty when the box crosses a zero meridian
            elif base[0] > 0:
                assert keep.sum() > 1000
        # ---- distance-range counts
        osh.set_deleted(gone)
        for base, unit, bounds in (((38.8951, 30.25), "km", [0.0, 200.0, 400.0, 600.0, 800.0]), ((52.52, 13.405), "miles", [100.0, 500.0, 1000.0])):
            for terms, qt, oop in cases:
                od, _, otot = osh.search_exhaustive(terms, oop, n_docs)
                q = sh.make_queries([terms], qt)
                counts, other, tot = sh.facet_count(q, off["loc"], "point", range_lower_bounds=bounds, base=base, unit=unit)
                b = np.searchsorted(np.asarray(bounds), O.geo_distances(v["loc"][od], base, unit), side="right") - 1
                assert tot == otot and other == int((b < 0).sum())
                assert np.array_equal(counts, np.bincount(b[b >= 0], minlength=len(bounds)))
        # ---- sort by distance
        for terms, qt, oop in cases:
            q = sh.make_queries([terms], qt)
            ad, as_, atot = osh.search_exhaustive(terms, oop, n_docs)
            for srt in ([("loc", False, bases[0])], [("loc", True, bases[1])], [("coarse", False, bases[0]), ("h", True, None)],
                        [("coarse", True, bases[1])]):
                for k, flt in ((10, None), (53, [(off["h"], "u16", 1000, 30000)])):
                    keep = np.ones(len(ad), bool) if flt is None else (v["h"][ad] >= 1000) & (v["h"][ad] < 30000)
                    md, ms = ad[keep], as_[keep]
                    cols = []
                    for name, desc, base in srt:
                        x = O.geo_distances(v[name][md], base, "sortkey") if base else v[name][md].astype(np.float64)
                        cols.append((-x if desc else x).tolist())
                    order = sorted(range(len(md)), key=lambda i: tuple(c[i] for c in cols) + (-float(ms[i]),))[:k]
                    spec = [(off[n], "point", d_, b_) if b_ else (off[n], "u16", d_) for n, d_, b_ in srt]
                    doc, score, tot = sh.search_lexical_sorted(q, spec, k, facet_filter=flt)
                    assert tot == len(md) and len(doc) == len(order), (terms, srt, k)
                    for name, _, base in srt:
                        assert np.array_equal(v[name][doc], v[name][md[order]]), (terms, srt, k, name)
                    assert np.allclose(score, ms[order], rtol=1e-4), (terms, srt, k)
        with pytest.raises(S.SeekStormHipError):
            sh.facet_count(q, off["loc"], "point", range_lower_bounds=[0.0], base=(0.0, 0.0), unit="km", facet_filter=[(off["coarse"] + 1, "point", (0.0, 0.0), 0.0, 1.0, "km")])
    finally:
        sh.set_deleted([])
        osh.set_deleted([])


def test_result_sort_by_a_string_facet(S, O, lex):
    """result_sort over a String16 facet (min_heap.rs:860-897): docs ordered by the STRING of their value id (byte-wise UTF-8,
    Rust's String order), ids with equal strings tie, then the next field / the score.  The host derives a rank column
    (Shard.string_facet_rank_column) and the device sorts by it; against a brute-force ordering by the strings themselves."""
    sh, osh, n_docs = lex
    rng = np.random.default_rng(41)
    words = ["zeta", "alpha", "Alpha", "beta", "\u00e9clair", "eclair", "omega", "beta", "a", "", "zz", "\u4e2d\u6587", "alpha ", "b", "B", "beta"]
    rec = np.dtype([("cat", "<u2"), ("h", "<u2")])
    v = np.zeros(n_docs, rec)
    v["cat"] = rng.integers(0, len(words), n_docs); v["h"] = rng.integers(0, 9, n_docs)
    raw, rank_off = S.Shard.string_facet_rank_column(v.view(np.uint8).reshape(n_docs, rec.itemsize), 0, "string16", words)
    assert raw.shape[1] == rec.itemsize + 4 and rank_off == rec.itemsize
    sh.upload_facets(raw)
    skey = [w.encode("utf-8") for w in words]
    for terms, qt, oop in (([10, 9, 8], S.QueryType.Union, O.OP_OR), ([10, 9], S.QueryType.Intersection, O.OP_AND)):
        q = sh.make_queries([terms], qt)
        ad, as_, atot = osh.search_exhaustive(terms, oop, n_docs)
        for desc, second in ((False, None), (True, None), (False, ("h", True))):
            for k in (10, 77):
                def key(i):
                    kb = skey[int(v["cat"][ad[i]])]
                    first = tuple(-b for b in kb) + (1,) if desc else tuple(kb) + (-1,)   # descending: reversed byte order, longer first
                    nxt = (-int(v["h"][ad[i]]),) if second else ()
                    return (first,) + nxt + (-float(as_[i]),)
                order = sorted(range(len(ad)), key=key)[:k]
                spec = [(rank_off, "u32", desc)] + ([(2, "u16", True)] if second else [])
                doc, score, tot = sh.search_lexical_sorted(q, spec, k)
                assert tot == atot and len(doc) == len(order)
                assert [skey[int(c)] for c in v["cat"][doc]] == [skey[int(c)] for c in v["cat"][ad[order]]], (terms, desc, k)
                if second:
                    assert np.array_equal(v["h"][doc], v["h"][ad[order]])
                assert np.allclose(score, as_[order], rtol=1e-4)


def test_index_two_shards_result_sort(S, O):
    """Index.search with result_sort over two shards (result_ordering_root, min_heap.rs:56-300): every shard's best under the
    sort, merged under the same order with each doc's facet values read from its own shard -- against the same sort on ONE
    shard holding the whole corpus (sort keys without ties decide the order alone: shard-local idf changes scores, not it)."""
    n_docs, S_n = 60_000, 2
    voc = [3000, 3600, 4000]
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = O.lex_corpus(n_docs, voc)
    rng = np.random.default_rng(3)
    rec = np.dtype([("u", "<u4"), ("g", "<f8"), ("loc", "<u8")])
    v = np.zeros(n_docs, rec)
    v["u"] = rng.permutation(n_docs); v["g"] = rng.standard_normal(n_docs)
    v["loc"] = O.morton_encode(rng.random(n_docs) * 50 + 10, rng.random(n_docs) * 60 + 5)
    raw = v.view(np.uint8).reshape(n_docs, rec.itemsize)
    whole = S.Shard(0)
    whole.upload_lexical(n_docs, dl, offs, docs, tfs)
    whole.upload_facets(raw)
    shards = []
    for sid in range(S_n):  # doc g -> shard g % S, local id g // S (index.rs:5284)
        sel = np.arange(sid, n_docs, S_n)
        o2, d2, t2 = [0], [], []
        for t in range(len(voc)):
            d = docs[int(offs[t]):int(offs[t + 1])]
            f = tfs[int(offs[t]):int(offs[t + 1])]
            m = (d % S_n) == sid
            d2.append(d[m] // S_n); t2.append(f[m]); o2.append(o2[-1] + int(m.sum()))
        sh = S.Shard(0, shard_id=sid)
        sh.upload_lexical(len(sel), dl[sel], np.asarray(o2, np.uint64), np.concatenate(d2).astype(np.uint32), np.concatenate(t2))
        sh.upload_facets(np.ascontiguousarray(raw[sel]))
        shards.append(sh)
    idx = S.Index(shards)
    base = (38.9, 30.2)
    for terms, qt in (([0, 1, 2], S.QueryType.Union), ([0, 1], S.QueryType.Intersection)):
        q = whole.make_queries([terms], qt)
        for sort in ([(0, "u32", True)], [(4, "f64", False)], [(12, "point", False, base)]):
            for off_, length, flt in ((0, 25, None), (7, 10, [(0, "u32", 5000, 50000)])):
                wd, ws, wtot = whole.search_lexical_sorted(q, sort, off_ + length, facet_filter=flt)
                ro = idx.search(terms, None, qt, S.SearchMode.Lexical, off_, length, strict=True, facet_filter=flt, result_sort=sort)
                assert ro.result_count_total == wtot
                assert [r.doc_id for r in ro.results] == [int(x) for x in wd[off_:off_ + length]], (terms, sort, off_, length)
    for sh in shards + [whole]:
        sh.close()
