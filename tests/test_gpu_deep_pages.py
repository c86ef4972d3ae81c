"""DEEP PAGES: offset + length beyond SS_MAX_K = 1024 results (search.rs:1658-1659 -- the crate's top_k is unbounded).

The kernels' top-k structures hold SS_MAX_K; a deeper page is answered behind the ABI in passes of SS_MAX_K results, every pass the
ordinary search under (what excluded docs before) | (the docs the earlier passes returned) -- csrc/ss_api.hip "deep pages".  Every
shape the library answers at k <= SS_MAX_K must therefore come back at any k with the oracle's answer: both tiers, NOT terms,
tombstones, a facet filter, several indexed fields (a union under a field filter composed from sub-queries), phrases; f32 and i8
vectors with several records per doc, tombstones and a threshold -- and a list that runs dry before the page is full.
"""
import numpy as np
import pytest

from test_gpu_shape_sweep import _check, _single_field_world, _fields_world, _gated_union_oracle

pytestmark = pytest.mark.gpu

REL = 1e-4


@pytest.fixture(scope="module")
def S():
    import seekstorm_amd
    return seekstorm_amd


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def _rows_well_formed(doc, score, cnt, k):
    n = int(cnt)
    assert np.all(doc[n:] == 0xFFFFFFFF) and np.all(score[n:] == 0)
    assert len(set(map(int, doc[:n]))) == n            # no doc twice: a pass never returns a doc of an earlier pass
    assert np.all(score[:n][:-1] >= score[:n][1:])     # one descending list across the passes' seams


@pytest.mark.parametrize("with_tier", [False, True])
def test_lexical_deep_pages_one_indexed_field(S, O, with_tier):
    sh, osh, n_docs, n_terms = _single_field_world(S, O, with_tier)
    # (terms, NOT terms, op): frequent terms 0..13 (70-80 % of 40 K docs), mid 14..37, rare 38..49 (the sparse tier when with_tier)
    cells = [([0], [], "or"), ([20], [], "or"), ([0, 1], [], "and"), ([0, 20], [], "or"), ([1, 16, 30], [], "or"), ([2, 3, 4], [15], "and"),
             ([14, 15, 16, 17, 18, 19], [], "or"), ([0, 1, 2, 3, 4, 5, 6, 7, 8], [20, 21], "or"), ([0, 40], [], "or"), ([5, 38, 45], [16], "or"),
             ([45], [], "or"), ([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11], [], "and"), ([3, 39], [], "and")]
    gone = list(range(5, n_docs, 89))
    q = sh.make_queries([c[0] for c in cells], [S.QueryType.Intersection if c[2] == "and" else S.QueryType.Union for c in cells], [c[1] for c in cells])
    for deleted in ((), gone):
        sh.set_deleted(deleted)
        osh.set_deleted(deleted)
        for k in (1025, 2048, 2500, 7000):
            for rt in (S.ResultType.TopkCount, S.ResultType.Topk):
                doc, score, cnt, tot = sh.search_lexical_batch(q, k, rt, reference_shortcuts=False)
                for i, (terms, nots, op) in enumerate(cells):
                    od, os_, otot = osh.search_exhaustive(terms, O.OP_AND if op == "and" else O.OP_OR, k, nots)
                    _check(doc[i], score[i], cnt[i], tot[i], od, os_, otot, rt, k, S, ("deep 1f", with_tier, terms, nots, op, k, rt, bool(deleted)))
                    _rows_well_formed(doc[i], score[i], cnt[i], k)
    # the page is a prefix of the deeper page (the order is total), and k = SS_MAX_K + 1 extends k = SS_MAX_K by exactly one result
    sh.set_deleted(())
    osh.set_deleted(())
    a = sh.search_lexical_batch(q, 1024, S.ResultType.TopkCount, reference_shortcuts=False)
    b = sh.search_lexical_batch(q, 1025, S.ResultType.TopkCount, reference_shortcuts=False)
    for i in range(len(cells)):
        n = int(a[2][i])
        assert np.array_equal(a[0][i][:n], b[0][i][:n]) and np.array_equal(a[1][i][:n], b[1][i][:n]) and a[3][i] == b[3][i]
        assert int(b[2][i]) == min(1025, int(b[3][i]))
    # Count needs no page at all
    c = sh.search_lexical_batch(q, 5000, S.ResultType.Count, reference_shortcuts=False)
    assert np.array_equal(c[3], b[3])
    # the mirror's single-query entry: a page at offset 3000
    ro = sh.search_lexical_shard([0, 20], S.QueryType.Union, 3000, 25)
    od, os_, otot = osh.search_exhaustive([0, 20], O.OP_OR, 3025, [])
    assert not ro.cpu_dispatch and ro.result_count_total == otot
    # (the shard task returns (0, offset + length); the planner cuts the page -- search.rs:1658-1659, 2098-2119)
    assert len(ro.results) == 25 and np.allclose([r.score for r in ro.results], os_[3000:], rtol=REL)
    sh.close()


def test_lexical_deep_page_under_a_facet_filter(S, O):
    sh, osh, n_docs, n_terms = _single_field_world(S, O, False)
    rng = np.random.default_rng(5)
    val = rng.integers(0, 100, n_docs).astype(np.uint8)
    sh.upload_facets(val.reshape(n_docs, 1))
    lo, hi = 10, 90
    ff = [(0, "u8", lo, hi)]  # passes iff lo <= value < hi
    keep = (val >= lo) & (val < hi)
    gone = np.nonzero(~keep)[0].tolist()
    osh.set_deleted(gone)  # the oracle's statement of the filter: a failing doc neither counts nor ranks
    q = sh.make_queries([[0, 20], [1, 2]], [S.QueryType.Union, S.QueryType.Intersection])
    for k in (1500, 3000):
        doc, score, cnt, tot = sh.search_lexical_batch(q, k, S.ResultType.TopkCount, reference_shortcuts=False, facet_filter=ff)
        for i, (terms, op) in enumerate((([0, 20], O.OP_OR), ([1, 2], O.OP_AND))):
            od, os_, otot = osh.search_exhaustive(terms, op, k, [])
            _check(doc[i], score[i], cnt[i], tot[i], od, os_, otot, S.ResultType.TopkCount, k, S, ("deep facet", terms, k))
            _rows_well_formed(doc[i], score[i], cnt[i], k)
            assert keep[doc[i][:int(cnt[i])]].all()
    sh.close()


def test_lexical_deep_pages_three_indexed_fields(S, O):
    n_docs, n_fields, boost = 30_000, 3, [2.0, 1.0, 0.5]
    dfs = [int(n_docs * x) for x in np.linspace(0.78, 0.66, 12)] + [int(n_docs * x) for x in np.geomspace(0.30, 0.01, 22)] + [int(x) for x in np.geomspace(300, 8, 8)]
    dl, offs, docs, fields, tfs = _fields_world(O, n_docs, n_fields, dfs, 88, lambda t: 0.03 if t < 12 else 0.4)
    nd = 34
    e = int(offs[nd])
    sh = S.Shard(0)
    sh.upload_lexical_fields(n_docs, dl, boost, offs[:nd + 1], docs[:e], fields[:e], tfs[:e])
    assert sh.append_sparse_fields(offs[nd:] - offs[nd], docs[e:], fields[e:], tfs[e:]) == nd
    ex = lambda terms, op, k, nots, deleted, filt=(): O.search_fields_exhaustive(n_docs, dl, boost, offs, docs, fields, tfs, terms, op, k, nots, deleted, field_filter=filt)[:3]
    gone = list(range(7, n_docs, 83))
    sh.set_deleted(gone)
    gone_set = set(gone)
    per_term = {}
    for t in range(len(dfs)):
        d, s_, _ = ex([t], O.OP_OR, n_docs, (), gone)
        a, b = int(offs[t]), int(offs[t + 1])
        per_term[t] = (dict(zip(d.tolist(), s_.tolist())), docs[a:b], fields[a:b])
    cells = [([0, 14], [], "or"), ([1, 2], [], "and"), ([3, 15, 36], [20], "or"), ([0, 1, 2, 3, 4, 5, 6, 7], [], "or"), ([13], [], "or")]
    for filt in ((), (0,), (1, 2)):
        q = sh.make_queries([c[0] for c in cells], [S.QueryType.Union if c[2] == "or" else S.QueryType.Intersection for c in cells], [c[1] for c in cells],
                            field_filter=filt)
        for k in (1200, 4100):
            for rt in (S.ResultType.TopkCount, S.ResultType.Topk):
                doc, score, cnt, tot = sh.search_lexical_batch(q, k, rt, reference_shortcuts=False)
                for i, (terms, nots, op) in enumerate(cells):
                    if op == "or" and filt and len(terms) > 1:  # composed from the reference's sub-queries (8 terms: 255 of them per pass)
                        od, os_, otot = _gated_union_oracle(per_term, terms, nots, filt, gone_set, k)
                    else:
                        od, os_, otot = ex(terms, O.OP_AND if op != "or" else O.OP_OR, k, nots, gone, filt if (op != "or" or len(terms) == 1) else ())
                    _check(doc[i], score[i], cnt[i], tot[i], od, os_, otot, rt, k, S, ("deep 3f", terms, nots, op, filt, k, rt))
                    _rows_well_formed(doc[i], score[i], cnt[i], k)
    sh.close()


def test_phrase_deep_page(S, O):
    from test_gpu_phrase import _corpus
    n_docs = 30_000
    dfs = [20_000, 19_000, 7_000, 300]
    dl, offs, docs, tfs, positions = _corpus(O, n_docs, dfs, 43, [([0, 1], 1500), ([0, 1, 2], 30)])
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, offs, docs, tfs, positions)
    osh = O.Shard(n_docs, dl, offs, docs, tfs)
    osh.set_positions(positions)
    q = sh.make_queries([[0, 1], [0, 1, 2]], S.QueryType.Phrase)
    for k in (1100, 2600):
        doc, score, cnt, tot = sh.search_lexical_batch(q, k, S.ResultType.TopkCount)
        for i, ph in enumerate(([0, 1], [0, 1, 2])):
            od, os_, otot = osh.search_phrase(ph, list(range(len(ph))), n_docs)
            assert i or len(od) > 1500  # the two-word phrase has more matches than two passes hold
            _check(doc[i], score[i], cnt[i], tot[i], od[:k], os_[:k], len(od), S.ResultType.TopkCount, k, S, ("deep phrase", ph, k))
            _rows_well_formed(doc[i], score[i], cnt[i], k)
    sh.close()


def _check_vec(doc, score, cnt, od, os_, k, abs_tol):
    n = int(cnt)
    assert n == len(od), (n, len(od))
    _rows_well_formed(doc, score, cnt, k)
    assert np.allclose(score[:n], os_, rtol=REL, atol=abs_tol)
    if n:
        band = abs(float(os_[-1])) * REL + abs_tol
        clear = lambda dd, ss: {int(x) for x, y in zip(dd, ss) if y > os_[-1] + 2 * band}
        assert clear(doc[:n], score[:n]) <= {int(x) for x in od} and clear(od, os_) <= {int(x) for x in doc[:n]}


@pytest.mark.parametrize("nq", [3, 70])
def test_vector_deep_pages_f32(S, O, nq):
    """f32 rows, several records per doc, tombstones; 70 queries = two groups of per-query exclusion bitmaps"""
    n_rows, dim = 24_000, 64
    rows = O.vec_gen(O.VEC_SEED, 0, n_rows, dim)
    rng = np.random.default_rng(3)
    ids = rng.integers(0, 9_000, n_rows).astype(np.uint32)  # ~ 2.7 records per doc, doc ids in no order
    qs = O.vec_gen(O.VECQ_SEED, 0, nq, dim)
    sh = S.Shard(0)
    sh.upload_vectors(rows, ids)
    n_distinct = len(set(ids.tolist()))
    for gone in ((), sorted(set(ids[::37].tolist()))):
        sh.set_deleted(list(gone))
        for k in ((1025, 3000) if nq == 3 else (2100,)):
            doc, score, cnt, tot = sh.search_vector_batch(qs, k)
            for i in range(0, nq, 1 if nq == 3 else 9):
                od, os_, _, _ = O.vec_search(rows, qs[i], k, row_doc_ids=ids, deleted=list(gone))
                _check_vec(doc[i], score[i], cnt[i], od, os_, k, 2e-6)
                assert not set(map(int, doc[i][:int(cnt[i])])) & set(gone)
        # a page deeper than the shard: every live doc once, then empty slots
        doc, score, cnt, tot = sh.search_vector_batch(qs[:2], 10_000)
        assert all(int(c) == n_distinct - len(gone) for c in cnt)
        for i in range(2):
            _rows_well_formed(doc[i], score[i], cnt[i], 10_000)
    sh.set_deleted([])
    # a threshold: the passes stop where the candidates do
    full = rows @ qs[0]
    best = {}
    for r in range(n_rows):
        best[int(ids[r])] = max(best.get(int(ids[r]), -2.0), float(full[r]))
    order = sorted(best.values(), reverse=True)
    from seekstorm_amd.search import SIMILARITY_NORMALIZATION_64_I8 as NORM
    t = (np.float32(0.5 * (order[1499] + order[1500])) * np.float32(NORM) + np.float32(1.0)) * np.float32(0.5)
    doc, score, cnt, tot = sh.search_vector_batch(qs[:1], 4000, similarity_threshold=float(t))
    from seekstorm_amd.search import threshold_raw
    want = int((np.asarray(order, np.float32) >= np.float32(threshold_raw(float(t)))).sum())  # `score < threshold -> reject` (vector.rs:423)
    assert 1400 < want < 1600 and abs(int(cnt[0]) - want) <= 3 and np.allclose(score[0][:1400], order[:1400], rtol=REL, atol=2e-6)
    _rows_well_formed(doc[0], score[0], cnt[0], 4000)
    sh.close()


def test_vector_deep_pages_i8(S, O):
    n_rows, dim, k = 9_000, 128, 2500
    rows = O.quantize_i8(O.vec_gen(O.VEC_SEED, 0, n_rows, dim))
    qs = O.quantize_i8(O.vec_gen(O.VECQ_SEED, 0, 4, dim))
    sh = S.Shard(0)
    sh.upload_vectors_i8(rows)
    gone = [3, 500, 8000]
    sh.set_deleted(gone)
    doc, score, cnt, tot = sh.search_vector_batch_i8(qs, k)
    for i in range(len(qs)):
        od, os_, _, _ = O.vec_search_i8(rows, qs[i], k, deleted=gone)
        assert cnt[i] == len(od) == k
        assert np.array_equal(score[i][:k], os_)  # integer dot products: bit-exact across the seams
        kth = os_[-1]
        assert {int(x) for x, y in zip(doc[i][:k], score[i]) if y > kth} == {int(x) for x, y in zip(od, os_) if y > kth}
        _rows_well_formed(doc[i], score[i], cnt[i], k)
    sh.close()


def test_sorted_deep_pages(S, O):
    """result_sort (min_heap.rs:574-1050) beyond SS_MAX_K results: the order (field 1, field 2, score desc, doc asc) is total, so a deep
    page sorted by facets is peeled like any other; with a facet filter and tombstones on top"""
    sh, osh, n_docs, n_terms = _single_field_world(S, O, False)
    rng = np.random.default_rng(8)
    rec = np.dtype([("a", "u1"), ("b", "<i2")])
    v = np.zeros(n_docs, rec)
    v["a"] = rng.integers(0, 12, n_docs)       # a dozen values: long tie groups, broken by the second field, then by the score
    v["b"] = rng.integers(-400, 400, n_docs)
    sh.upload_facets(v.view(np.uint8).reshape(n_docs, 3))
    gone = list(range(3, n_docs, 97))
    sh.set_deleted(gone)
    osh.set_deleted(gone)
    cases = [([0, 20], O.OP_OR, S.QueryType.Union), ([1, 2], O.OP_AND, S.QueryType.Intersection), ([25], O.OP_OR, S.QueryType.Union)]
    for terms, oop, qt in cases:
        q = sh.make_queries([terms], qt)
        md, ms, mtot = osh.search_exhaustive(terms, oop, n_docs, [])  # every match with its score
        for spec, flt in (([(0, "u8", False), (1, "i16", True)], None), ([(1, "i16", False)], [(0, "u8", 2, 9)])):
            keep = np.ones(len(md), bool) if flt is None else (v["a"][md] >= 2) & (v["a"][md] < 9)
            d_, s_ = md[keep], ms[keep]
            keys = [d_.astype(np.int64), -s_.astype(np.float64)]  # lexsort: last key first
            for off_, ty, desc in reversed(spec):
                col = v["a" if off_ == 0 else "b"][d_].astype(np.int64)
                keys.append(-col if desc else col)
            order = np.lexsort(keys)
            for k in (1030, 2600):
                doc, score, tot = sh.search_lexical_sorted(q, spec, k, facet_filter=flt)
                n = min(k, len(d_))
                assert tot == len(d_) and len(doc) == n, (terms, spec, k, tot, len(d_), len(doc))
                assert len(set(doc.tolist())) == n
                want_d, want_s = d_[order][:n], s_[order][:n]
                # the sort fields agree position by position; docs may swap only where sort fields AND scores (within tolerance) tie
                for off_, ty, desc in spec:
                    name = "a" if off_ == 0 else "b"
                    assert np.array_equal(v[name][doc], v[name][want_d]), (terms, spec, k, name)
                assert np.allclose(score, want_s, rtol=REL, atol=1e-7)
                assert (doc == want_d).mean() > 0.98
    sh.close()


def test_hybrid_deep_page_and_concurrent_callers(S, O):
    """a deep HYBRID page through the mirror's planner (two deep shard searches + RRF on the host, search.rs:1962-2035) over two shards,
    and deep pages issued while other threads search the same shard through the coalescer: the passes swap the shard's exclusion bitmap
    under the shard lock -- nobody else may ever see it"""
    import threading
    n_docs, voc, dim = 60_000, list(range(2600, 4096, 150)), 64
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = O.lex_corpus(n_docs, voc)
    rows = O.vec_gen(O.VEC_SEED, 0, n_docs, dim)
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, offs, docs, tfs)
    sh.upload_vectors(rows)
    qs = O.vec_gen(O.VECQ_SEED, 0, 2, dim)
    ix = S.Index([sh])
    terms = [3, 7, 9]
    deep = ix.search(terms, qs[0], S.QueryType.Union, S.SearchMode.Hybrid, 2000, 30, normalize_query=False)
    # the same page from the two deep lists fused here: RRF score of a doc = sum over the lists holding it of 1 / (0.6 + rank)
    ld, ls, lc, lt = sh.search_lexical_batch(sh.make_queries([terms], S.QueryType.Union), 2030, reference_shortcuts=False)
    vd, vs, vc, vt = sh.search_vector_batch(qs[:1], 2030)
    assert int(lc[0]) == 2030 and int(vc[0]) == 2030
    fused = {}
    for lst in (ld[0], vd[0]):
        for r, d in enumerate(lst.tolist()):
            fused[d] = np.float32(fused.get(d, np.float32(0)) + np.float32(1.0) / (np.float32(0.6) + np.float32(r)))
    want = sorted(fused.items(), key=lambda e: (-float(e[1]), e[0]))[2000:2030]
    assert not deep.cpu_dispatch and len(deep.results) == 30
    assert np.allclose([r.score for r in deep.results], [float(s_) for _, s_ in want], rtol=1e-6)
    assert [r.doc_id for r in deep.results] == [d for d, _ in want]
    # ---- concurrency: two threads page deep, four threads ask ordinary single queries of the same shard
    q1 = sh.make_queries([[3, 7, 9]], S.QueryType.Union)
    q2 = sh.make_queries([[5, 2]], S.QueryType.Union)
    ref_deep = sh.search_lexical_batch(q1, 3000, reference_shortcuts=False)
    ref_vdeep = sh.search_vector_batch(qs[1:2], 2500)
    ref_small = [sh.search_lexical_batch(q, 10, reference_shortcuts=False) for q in (q1, q2)]
    ref_vsmall = sh.search_vector_batch(qs[:1], 20)
    bad, stop = [], threading.Event()

    def deep_lex():
        for _ in range(12):
            got = sh.search_lexical_batch(q1, 3000, reference_shortcuts=False)
            if not all(np.array_equal(a, b) for a, b in zip(got, ref_deep)):
                bad.append("deep lexical")

    def deep_vec():
        for _ in range(6):
            got = sh.search_vector_batch(qs[1:2], 2500)
            if not all(np.array_equal(a, b) for a, b in zip(got[:3], ref_vdeep[:3])):
                bad.append("deep vector")

    def small(i):
        while not stop.is_set():
            if i == 3:
                got = sh.search_vector_batch(qs[:1], 20)
                if not all(np.array_equal(a, b) for a, b in zip(got[:3], ref_vsmall[:3])):
                    bad.append("small vector")
            else:
                got = sh.search_lexical_batch((q1, q2)[i & 1], 10, reference_shortcuts=False)
                if not all(np.array_equal(a, b) for a, b in zip(got, ref_small[i & 1])):
                    bad.append("small lexical")

    th = [threading.Thread(target=small, args=(i,)) for i in range(4)]
    dth = [threading.Thread(target=deep_lex), threading.Thread(target=deep_vec)]
    for t in th + dth:
        t.start()
    for t in dth:
        t.join()
    stop.set()
    for t in th:
        t.join()
    assert not bad, sorted(set(bad))
    sh.close()


def test_device_pointer_entries_page_deep_too(S, O):
    """ss_bm25_search_dev / ss_vec_search_dev with k > SS_MAX_K: the passes are steered from the host (one trip of the queries there, a
    synchronous call), the page is left in the caller's DEVICE arrays -- the same page as the host-pointer entry's"""
    import torch
    from seekstorm_amd import _native as N
    sh, osh, n_docs, n_terms = _single_field_world(S, O, True)
    dev = torch.device("cuda", 0)
    cells = [[0, 20], [1, 16, 30], [5, 38, 45], [45]]
    q = sh.make_queries(cells, S.QueryType.Union)
    nq, k = len(q), 2500
    want = sh.search_lexical_batch(q, k, S.ResultType.TopkCount, reference_shortcuts=False)
    qd = torch.from_numpy(q.view(np.uint8).reshape(nq, -1).copy()).to(dev)
    doc = torch.full((nq, k), -1, dtype=torch.int32, device=dev); score = torch.zeros((nq, k), dtype=torch.float32, device=dev)
    cnt = torch.zeros((nq,), dtype=torch.int32, device=dev); tot = torch.zeros((nq,), dtype=torch.int64, device=dev)
    st = torch.cuda.Stream(device=dev)
    ops = 2 | (3 << 8) | (3 << 16) | (1 << 28)  # unions, <= 3 terms, may name sparse-tier terms
    N.check(N.lib().ss_bm25_search_dev(sh._h, nq, qd.data_ptr(), k, int(S.ResultType.TopkCount), ops, doc.data_ptr(), score.data_ptr(), cnt.data_ptr(),
                                       tot.data_ptr(), st.cuda_stream), "ss_bm25_search_dev")
    st.synchronize()
    assert np.array_equal(doc.cpu().numpy().view(np.uint32), want[0]) and np.array_equal(score.cpu().numpy(), want[1])
    assert np.array_equal(cnt.cpu().numpy().view(np.uint32), want[2]) and np.array_equal(tot.cpu().numpy().view(np.uint64), want[3])
    assert int(want[2].max()) == k
    sh.close()
    # vectors
    n_rows, dim, kv = 12_000, 64, 2100
    rows = O.vec_gen(O.VEC_SEED, 0, n_rows, dim)
    qs = O.vec_gen(O.VECQ_SEED, 0, 3, dim)
    sv = S.Shard(0)
    sv.upload_vectors(rows)
    sv.set_deleted([5, 77, 4000])
    wd, ws, wc, wt = sv.search_vector_batch(qs, kv)
    tq = torch.from_numpy(qs).to(dev)
    vdoc = torch.full((3, kv), -1, dtype=torch.int32, device=dev); vscore = torch.zeros((3, kv), dtype=torch.float32, device=dev)
    vcnt = torch.zeros((3,), dtype=torch.int32, device=dev); vtot = torch.zeros((3,), dtype=torch.int64, device=dev)
    N.check(N.lib().ss_vec_search_dev(sv._h, 3, tq.data_ptr(), kv, N.FLT_MIN_NEG, vdoc.data_ptr(), vscore.data_ptr(), vcnt.data_ptr(), vtot.data_ptr(),
                                      st.cuda_stream), "ss_vec_search_dev")
    st.synchronize()
    assert np.array_equal(vdoc.cpu().numpy().view(np.uint32), wd) and np.array_equal(vscore.cpu().numpy(), ws)
    assert np.array_equal(vcnt.cpu().numpy().view(np.uint32), wc) and int(wc.min()) == kv
    sv.close()


def test_vector_deep_pages_under_ann_modes(S, O):
    """Nprobe / NprobeSimilaritythreshold with k > SS_MAX_K (vector.rs:1300-1392): the passes re-select the same clusters, pass 0 reports
    them; integer dots are exact, so the whole page is the oracle's -- including the queries whose clusters hold fewer records than k"""
    from test_gpu_ann import clustered, queries_near
    lc = [6, 9]
    rows32, child = clustered(O, 77, lc, 64, lo=500, hi=1400)
    rows = O.quantize_i8(rows32)
    qs = O.quantize_i8(queries_near(O, rows32, 78, 6))
    sh = S.Shard(0)
    sh.upload_vectors_i8(rows)
    sh.set_clusters(lc, child)
    gone = [int(x) for x in range(7, len(rows), 301)]
    sh.set_deleted(gone)
    k = 2600
    for am, kw in ((S.AnnMode.Nprobe(2), dict(n_probe=2)), (S.AnnMode.Nprobe(4), dict(n_probe=4))):
        doc, score, cnt, tot, ncl = sh.search_vector_batch_i8(qs, k, ann_mode=am, with_clusters=True)
        deep_seen = 0
        for i in range(len(qs)):
            od, os_, otot, oobs, oncl = O.vec_search_i8_ann(rows, qs[i], k, lc, child, deleted=gone, **kw)
            assert ncl[i] == oncl and cnt[i] == len(od), (i, int(ncl[i]), oncl, int(cnt[i]), len(od))
            assert np.array_equal(score[i][:cnt[i]], os_)
            if len(od):
                kth = os_[-1]
                assert {int(x) for x, y in zip(doc[i][:cnt[i]], score[i]) if y > kth} == {int(x) for x, y in zip(od, os_) if y > kth}
            _rows_well_formed(doc[i], score[i], cnt[i], k)
            assert not set(map(int, doc[i][:int(cnt[i])])) & set(gone)
            deep_seen += int(cnt[i]) > 1024
        assert deep_seen >= 3  # (pages that really took more than one pass)
    sh.close()


def test_vector_deep_page_euclidean_f32(S, O):
    """f32 Euclidean: the scan ranks by the MFMA form of the distance and the kept records are rescored in the reference's summation order
    (vec_rescore_euclid_kernel) -- pass by pass; the page must still be one descending list and meet the oracle's within the tolerance"""
    n_rows, dim, k = 9_000, 96, 2300
    rows = O.vec_gen(O.VEC_SEED, 0, n_rows, dim, normalize=False) * np.float32(40.0)
    qs = O.vec_gen(O.VECQ_SEED, 0, 3, dim, normalize=False) * np.float32(40.0)
    sh = S.Shard(0)
    sh.set_vector_similarity("euclidean")
    sh.upload_vectors(rows)
    doc, score, cnt, tot = sh.search_vector_batch(qs, k)
    for i in range(3):
        od, os_, *_ = O.vec_search_euclid(rows, qs[i], k, simd_order=True)
        assert cnt[i] == len(od) == k
        _check_vec(doc[i], score[i], cnt[i], od, os_, k, 1e-5 * 1600)
    sh.close()
