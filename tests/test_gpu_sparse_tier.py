"""Images with a sparse tier (rare terms as plain sorted lists): unions / intersections / NOT terms / phrases naming sparse terms, several indexed fields, the device-pointer entry point, mixed phrase + plain batches."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import seekstorm_amd
    return seekstorm_amd


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def both(S, O):
    """one shard holding a lexical and a vector image over the same doc ids"""
    n_docs, voc, dim = 60_000, list(range(2600, 4096, 150)), 96
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = O.lex_corpus(n_docs, voc)
    rows = O.vec_gen(O.VEC_SEED, 0, n_docs, dim)
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, offs, docs, tfs)
    sh.upload_vectors(rows)
    yield sh, rows, n_docs, dim
    sh.close()


def _dense_corpus(O, n_docs, dfs, seed=77):
    """posting lists with the given document frequencies (fractions of n_docs), tf geometric, ascending docs"""
    rng = np.random.default_rng(seed)
    offs, docs, tfs = [0], [], []
    for df in dfs:
        d = np.sort(rng.choice(n_docs, int(df * n_docs), replace=False)).astype(np.uint32)
        docs.append(d)
        tfs.append(np.minimum(rng.geometric(0.6, len(d)), 60).astype(np.uint16))
        offs.append(offs[-1] + len(d))
    return np.asarray(offs, np.uint64), np.concatenate(docs), np.concatenate(tfs)


REL = 1e-4


VOC = [0, 1500, 2500, 3000, 3300, 3600, 3800, 3900, 4000, 4050, 4095]  # df from 0.05 % to 20 %


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


def _check_topk(doc, score, cnt, od, os_, abs_tol=0.0):
    """rows sorted desc; scores within REL of the oracle's; identical id sets outside the tie band of the k-th"""
    n = int(cnt)
    assert n == len(od)
    d, s = doc[:n], score[:n]
    assert np.all(s[:-1] >= s[1:])
    assert np.all(doc[n:] == 0xFFFFFFFF)
    assert len(set(map(int, d))) == n
    assert np.allclose(s, os_, rtol=REL, atol=abs_tol)
    if n:
        band = abs(float(os_[-1])) * REL + abs_tol
        clear = lambda dd, ss: {int(x) for x, y in zip(dd, ss) if y > os_[-1] + 2 * band}
        assert clear(d, s) <= {int(x) for x in od} and clear(od, os_) <= {int(x) for x in d}


def _same(a, b, what):
    for x, y, name in zip(a, b, ("doc", "score", "count", "total")):
        assert np.array_equal(x, y), (what, name)


def _oracle_check(S, O, osh, cs, oop, rt, got, k=10):
    doc, score, cnt, tot = got
    for i, (pos, neg) in enumerate(cs):
        od, os_, otot = osh.search_exhaustive(pos, oop, k, not_terms=neg)
        if rt != S.ResultType.Topk:
            assert int(tot[i]) == otot, (pos, neg, rt)
        if rt != S.ResultType.Count:
            _check_topk(doc[i], score[i], cnt[i], od, os_)


def _level_slices(n_docs, offs, docs, tfs, n_terms=None):
    """CSR of a corpus -> per 65 536-doc level (doclen slice bounds, offs, docs, tfs) over the first n_terms terms"""
    nt = len(offs) - 1 if n_terms is None else n_terms
    out = []
    for lv in range((n_docs + 65535) // 65536):
        lo, hi = lv * 65536, min(n_docs, (lv + 1) * 65536)
        lo_, do_, to_ = [0], [], []
        for t in range(nt):
            a, b = int(offs[t]), int(offs[t + 1])
            i0, i1 = a + int(np.searchsorted(docs[a:b], lo)), a + int(np.searchsorted(docs[a:b], hi))
            do_.append(docs[i0:i1]); to_.append(tfs[i0:i1]); lo_.append(lo_[-1] + (i1 - i0))
        out.append((lo, hi, np.asarray(lo_, np.uint64), np.concatenate(do_) if do_ else np.zeros(0, np.uint32),
                    np.concatenate(to_) if to_ else np.zeros(0, np.uint16)))
    return out


def _tiered_shard(S, O, n_docs=150_000, seed=21):
    """a dense image of 5 lists + 9 sparse lists that overlap each other and the dense lists; the oracle holds all 14 as ordinary lists"""
    rng = np.random.default_rng(seed)
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = [0], [], []
    for df in (0.004, 0.02, 0.07, 0.15, 0.33):
        d = np.sort(rng.choice(n_docs, int(df * n_docs), replace=False)).astype(np.uint32)
        docs.append(d); tfs.append(np.minimum(rng.geometric(0.6, len(d)), 60).astype(np.uint16)); offs.append(offs[-1] + len(d))
    nd = len(offs) - 1
    hot = np.sort(rng.choice(n_docs, 6000, replace=False))
    sp_n = [2, 50, 400, 1500, 3000, 9, 65, 2200, 700]
    s_offs, s_docs, s_tfs = [0], [], []
    for n in sp_n:
        d = np.sort(rng.choice(hot, n, replace=False)).astype(np.uint32)
        s_docs.append(d); s_tfs.append(np.minimum(rng.geometric(0.5, n), 30).astype(np.uint16)); s_offs.append(s_offs[-1] + n)
    d_offs, d_docs, d_tfs = np.asarray(offs, np.uint64), np.concatenate(docs), np.concatenate(tfs)
    s_offs, s_docs, s_tfs = np.asarray(s_offs, np.uint64), np.concatenate(s_docs), np.concatenate(s_tfs)
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, d_offs, d_docs, d_tfs)
    assert sh.append_sparse(s_offs, s_docs, s_tfs) == nd
    osh = O.Shard(n_docs, dl, np.concatenate([d_offs, d_offs[-1] + s_offs[1:]]), np.concatenate([d_docs, s_docs]), np.concatenate([d_tfs, s_tfs]))
    return sh, osh, nd, len(sp_n), hot, n_docs


def _check_against(osh, O, S, out, cases, op, k, rt):
    d, s_, c, t = out
    for i, (terms, nots) in enumerate(cases):
        od, os_, otot = osh.search_exhaustive(terms, op, k, not_terms=nots)
        assert int(t[i]) == otot, (op, k, i, terms, nots, int(t[i]), otot)
        if rt == S.ResultType.Count:
            continue
        assert c[i] == len(od), (op, k, i, terms, nots, int(c[i]), len(od))
        assert np.allclose(s_[i, :c[i]], os_, rtol=1e-4), (op, k, i, terms, nots)
        if len(od) < k:
            assert set(d[i, :c[i]].tolist()) == set(int(x) for x in od)


def _same_answers(a, b):
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_sparse_tier_queries_against_the_oracle(S, O):
    """rare terms in the SPARSE tier (plain sorted lists, no directory / probe rows; ss_bm25_append_sparse), queries mixing them with
    dense terms: unions (dense part through the ordinary kernels + every doc of a sparse list scored in full + per-query merge),
    intersections (the shortest sparse list drives), exact counts, NOT terms, tombstones, two appends, the coalesced single-query
    path -- against the oracle holding ALL lists as ordinary lists"""
    from seekstorm_amd import _native as N
    n_docs = 200_000
    dense_df = [0.002, 0.01, 0.04, 0.11, 0.3]
    sparse_n = [1, 3, 40, 250, 900, 1800, 7, 64, 65, 1200]
    dl = O.lex_doclen(n_docs)
    d_offs, d_docs, d_tfs = _dense_corpus(O, n_docs, dense_df, seed=5)
    rng = np.random.default_rng(11)
    s_offs, s_docs, s_tfs = [0], [], []
    hot = np.sort(rng.choice(n_docs, 3000, replace=False))  # sparse lists overlap each other and the dense lists often
    for n in sparse_n:
        d = np.sort(rng.choice(hot, n, replace=False)).astype(np.uint32)
        s_docs.append(d); s_tfs.append(np.minimum(rng.geometric(0.5, n), 30).astype(np.uint16)); s_offs.append(s_offs[-1] + n)
    s_docs, s_tfs, s_offs = np.concatenate(s_docs), np.concatenate(s_tfs), np.asarray(s_offs, np.uint64)
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, d_offs, d_docs, d_tfs)
    nd = len(dense_df)
    first = sh.append_sparse(s_offs[:7], s_docs[:int(s_offs[6])], s_tfs[:int(s_offs[6])])          # two appends: ids continue
    second = sh.append_sparse(s_offs[6:] - s_offs[6], s_docs[int(s_offs[6]):], s_tfs[int(s_offs[6]):])
    assert first == nd and second == nd + 6 and sh.sparse_info()[:2] == (len(sparse_n), int(s_offs[-1]))
    # the oracle: one shard with every list as an ordinary list (term ids: dense, then sparse)
    a_offs = np.concatenate([d_offs, d_offs[-1] + s_offs[1:]])
    osh = O.Shard(n_docs, dl, a_offs, np.concatenate([d_docs, s_docs]), np.concatenate([d_tfs, s_tfs]))
    assert [int(x) for x in sh.posting_count(list(range(nd, nd + len(sparse_n))))] == sparse_n
    nt_all = nd + len(sparse_n)
    cases = []
    for _ in range(60):
        n = int(rng.integers(1, 5))
        terms = [int(x) for x in rng.choice(nt_all, n, replace=False)]
        cases.append((terms, []))
    cases += [([nd + 3, nd + 4], []), ([nd + 5], []), ([0, nd], []), ([nd + 9, 4, 3], [2]), ([nd + 4, nd + 5, 1], [0])]
    for op, qt in ((O.OP_OR, S.QueryType.Union), (O.OP_AND, S.QueryType.Intersection)):
        tl = [c[0] for c in cases]
        nl = [c[1] for c in cases]
        q = sh.make_queries(tl, qt, nl)
        for k in (10, 100):
            for rt in (S.ResultType.TopkCount, S.ResultType.Count):
                d, s_, c, t = sh.search_lexical_batch(q, k, rt, reference_shortcuts=False)
                for i, (terms, nots) in enumerate(cases):
                    od, os_, otot = osh.search_exhaustive(terms, op, k, not_terms=nots)
                    assert int(t[i]) == otot, (op, k, i, terms, nots, int(t[i]), otot)
                    if rt == S.ResultType.Count:
                        continue
                    assert c[i] == len(od), (op, k, i, terms, c[i], len(od))
                    assert np.allclose(s_[i, :c[i]], os_, rtol=1e-4), (op, k, i, terms)
                    if len(od) < k:
                        assert set(d[i, :c[i]].tolist()) == set(int(x) for x in od)
    # a sparse NOT list inside an intersection (in a union: tests/test_gpu_round4.py)
    q = sh.make_queries([[3, nd + 8]], S.QueryType.Intersection, [[nd + 4]])
    d, s_, c, t = sh.search_lexical_batch(q, 10, reference_shortcuts=False)
    od, os_, otot = osh.search_exhaustive([3, nd + 8], O.OP_AND, 10, not_terms=[nd + 4])
    assert int(t[0]) == otot and np.allclose(s_[0, :c[0]], os_, rtol=1e-4)
    # tombstones: neither counted nor ranked, in either part
    gone = [int(x) for x in hot[::3]]
    sh.set_deleted(gone)
    osh.set_deleted(gone)
    q = sh.make_queries([[nd + 5, 3], [nd + 4, nd + 9, 2], [nd + 5, nd + 9]], [S.QueryType.Union, S.QueryType.Union, S.QueryType.Intersection])
    d, s_, c, t = sh.search_lexical_batch(q, 10, reference_shortcuts=False)
    for i, (terms, op) in enumerate((([nd + 5, 3], O.OP_OR), ([nd + 4, nd + 9, 2], O.OP_OR), ([nd + 5, nd + 9], O.OP_AND))):
        od, os_, otot = osh.search_exhaustive(terms, op, 10)
        assert c[i] == len(od) and np.allclose(s_[i, :c[i]], os_, rtol=1e-4)
        if op == O.OP_AND or True:
            assert int(t[i]) == otot, (i, int(t[i]), otot)
    sh.close()


def test_mixed_phrase_and_plain_batches_through_the_abi(S, O):
    """one C-ABI batch mixing phrase queries with unions and intersections (the coalescer merges whatever concurrent callers bring):
    the library runs it as two sub-batches and puts the answers back in the callers' order -- equal to the homogeneous calls"""
    n_docs = 30_000
    dl = O.lex_doclen(n_docs)
    rng = np.random.default_rng(8)
    offs, docs, tfs, pos = [0], [], [], []
    for df in (6000, 5000, 4000, 900):
        d = np.sort(rng.choice(n_docs, df, replace=False)).astype(np.uint32)
        t = np.minimum(rng.geometric(0.5, df), 6).astype(np.uint16)
        for n in t:
            pos.append(np.sort(rng.choice(40, int(n), replace=False)).astype(np.uint16))
        docs.append(d); tfs.append(t); offs.append(offs[-1] + df)
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, np.asarray(offs, np.uint64), np.concatenate(docs), np.concatenate(tfs), positions=np.concatenate(pos))
    tl = [[0, 1], [0, 1], [2, 1, 0], [3], [1, 2], [0, 2]]
    qt = [S.QueryType.Phrase, S.QueryType.Union, S.QueryType.Phrase, S.QueryType.Union, S.QueryType.Intersection, S.QueryType.Phrase]
    q = sh.make_queries(tl, qt)
    for rt in (S.ResultType.TopkCount, S.ResultType.Topk):
        mixed = sh.search_lexical_batch(q, 10, rt, reference_shortcuts=False)
        for i in range(len(tl)):
            alone = sh.search_lexical_batch(sh.make_queries([tl[i]], [qt[i]]), 10, rt, reference_shortcuts=False)
            for a, b in zip(mixed, alone):
                assert np.array_equal(a[i], b[0]), (i, int(rt))
    sh.close()


def test_sparse_tier_not_terms_in_unions_large_k_and_facet_filters(S, O):
    """the sparse tier next to the dense tier's abilities: a UNION that excludes a sparse term (alone and inside a batch, with dense NOT
    terms beside it, on a shard with tombstones), k up to SS_MAX_K, facet filters over queries naming sparse terms -- against the
    oracle holding every list as an ordinary list"""
    from seekstorm_amd import _native as N
    sh, osh, nd, ns, hot, n_docs = _tiered_shard(S, O)
    rng = np.random.default_rng(4)
    U, A = S.QueryType.Union, S.QueryType.Intersection
    # unions with sparse NOT terms: single calls, then a batch mixing them with ordinary queries
    cases = [([3, nd + 3], [nd + 4]), ([4, 2], [nd + 4]), ([nd + 7, nd + 2, 1], [nd + 3, 0]), ([2], [nd + 4, nd + 7]), ([nd + 4], [nd + 7]),
             ([4, 3, 2], [nd + 4, 1, nd + 8])]
    for strat in (N.BM25_AUTO, N.BM25_EXHAUSTIVE):
        sh.set_strategy(strat)
        for gone in ([], [int(x) for x in hot[::4]]):
            sh.set_deleted(gone); osh.set_deleted(gone)
            for k in (10, 100):
                for rt in (S.ResultType.TopkCount, S.ResultType.Topk, S.ResultType.Count):
                    for terms, nots in cases:
                        out = sh.search_lexical_batch(sh.make_queries([terms], U, [nots]), k, rt, reference_shortcuts=False)
                        if rt != S.ResultType.Topk:
                            _check_against(osh, O, S, out, [(terms, nots)], O.OP_OR, k, rt)
                        else:
                            od, os_, _ = osh.search_exhaustive(terms, O.OP_OR, k, not_terms=nots)
                            assert out[2][0] == len(od) and np.allclose(out[1][0, :len(od)], os_, rtol=1e-4)
                    mixed = [([0, 1, 2], []), cases[0], ([nd + 4, 3], []), cases[2], ([4], [2]), cases[3], ([1, nd + 6], [3])]
                    if rt != S.ResultType.Topk:
                        out = sh.search_lexical_batch(sh.make_queries([c[0] for c in mixed], U, [c[1] for c in mixed]), k, rt, reference_shortcuts=False)
                        _check_against(osh, O, S, out, mixed, O.OP_OR, k, rt)
    sh.set_strategy(N.BM25_AUTO)
    sh.set_deleted([]); osh.set_deleted([])
    # k beyond 128: the lists of both parts are merged at any k the dense tier serves
    big = [([nd + 4, 3], []), ([nd + 3, nd + 7, 2], []), ([4, nd + 8], [1]), ([nd + 4], []), ([nd + 4, nd + 3], []), ([0, 1], [])]
    for k in (200, 256, 700, 1024):
        for op, qt in ((O.OP_OR, U), (O.OP_AND, A)):
            out = sh.search_lexical_batch(sh.make_queries([c[0] for c in big], qt, [c[1] for c in big]), k, S.ResultType.TopkCount, reference_shortcuts=False)
            _check_against(osh, O, S, out, big, op, k, S.ResultType.TopkCount)
    # facet filters: the sparse kernel honours the filter's bitmap like the dense ones
    val = rng.integers(0, 100, n_docs).astype(np.uint8)
    sh.upload_facets(val.reshape(n_docs, 1))
    keep = (val >= 20) & (val < 70)
    gone = [int(x) for x in hot[1::5]]
    sh.set_deleted(gone)
    osh.set_deleted(sorted(set(np.nonzero(~keep)[0].tolist()) | set(gone)))
    fcases = [([nd + 4, 3], []), ([nd + 3, nd + 7, 2], [1]), ([nd + 4, nd + 7], []), ([2, 3], [nd + 4]), ([nd + 2], []), ([4, nd + 4], [nd + 7, 0])]
    for op, qt in ((O.OP_OR, U), (O.OP_AND, A)):
        for rt in (S.ResultType.TopkCount, S.ResultType.Count):
            out = sh.search_lexical_batch(sh.make_queries([c[0] for c in fcases], qt, [c[1] for c in fcases]), 10, rt, reference_shortcuts=False,
                                          facet_filter=[(0, "u8", 20, 70)])
            _check_against(osh, O, S, out, fcases, op, 10, rt)
            if rt != S.ResultType.Count:
                assert all(keep[int(x)] for i in range(len(fcases)) for x in out[0][i, :out[2][i]])
    # the next unfiltered call sees the tombstones only
    osh.set_deleted(gone)
    out = sh.search_lexical_batch(sh.make_queries([[nd + 4, 3]], U), 10, reference_shortcuts=False)
    _check_against(osh, O, S, out, [([nd + 4, 3], [])], O.OP_OR, 10, S.ResultType.TopkCount)
    sh.close()


def test_sparse_terms_through_the_device_pointer_entry_point(S, O):
    """ss_bm25_search_dev with ops_mask bit 28: a device-resident batch naming sparse terms is split on the host (one round trip) and
    answered into the caller's device arrays; without the bit a sparse term id is outside the dense vocabulary and flagged"""
    import torch
    from seekstorm_amd import _native as N
    sh, osh, nd, ns, hot, n_docs = _tiered_shard(S, O, n_docs=80_000, seed=9)
    cases = [([nd + 4, 3], []), ([0, 1, 2], []), ([nd + 3, nd + 7, 2], [1]), ([2, 3], [nd + 4]), ([nd + 1], [])]
    q = sh.make_queries([c[0] for c in cases], S.QueryType.Union, [c[1] for c in cases])
    nq, k = len(cases), 10
    qd = torch.from_numpy(q.view(np.uint8).copy()).cuda()
    doc = torch.zeros((nq, k), dtype=torch.int32, device="cuda"); score = torch.zeros((nq, k), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(nq, dtype=torch.int32, device="cuda"); tot = torch.zeros(nq, dtype=torch.int64, device="cuda")
    st = torch.cuda.Stream()
    for stream in (None, C.c_void_p(st.cuda_stream)):
        ops = 2 | (4 << 8) | (3 << 16) | (1 << 28)
        N.check(N.lib().ss_bm25_search_dev(sh._h, nq, qd.data_ptr(), k, N.RT_TOPKCOUNT, ops, doc.data_ptr(), score.data_ptr(), cnt.data_ptr(),
                                           tot.data_ptr(), stream), "ss_bm25_search_dev")
        torch.cuda.synchronize()
        out = (doc.cpu().numpy().view(np.uint32), score.cpu().numpy(), cnt.cpu().numpy().view(np.uint32), tot.cpu().numpy().view(np.uint64))
        _check_against(osh, O, S, out, cases, O.OP_OR, k, S.ResultType.TopkCount)
        doc.zero_(); score.zero_(); cnt.zero_(); tot.zero_()
    N.check(N.lib().ss_bm25_search_dev(sh._h, nq, qd.data_ptr(), k, N.RT_TOPKCOUNT, 2 | (4 << 8) | (3 << 16), doc.data_ptr(), score.data_ptr(),
                                       cnt.data_ptr(), tot.data_ptr(), None), "ss_bm25_search_dev")
    torch.cuda.synchronize()
    flagged = cnt.cpu().numpy().view(np.uint32) == 0xFFFFFFFF
    assert flagged.tolist() == [True, False, True, True, True]
    sh.close()


def test_sparse_tier_on_an_image_with_several_indexed_fields(S, O):
    """rare terms of a multi-field (BM25F) index in the sparse tier: their MERGED lists (every doc once, weight = the boosted sum over
    the doc's fields) next to the dense image's merged lists -- unions, intersections, NOT terms of either tier, tombstones, counts,
    against the brute-force BM25F oracle over ALL entries; field filters over sparse terms: intersections and single terms by the
    postings' field masks, unions of several terms composed from the reference's own sub-queries"""
    from seekstorm_amd import _native as N
    from test_gpu_parity import _fields_corpus, _check_topk
    n_docs, n_fields, boost = 100_000, 3, [2.0, 1.0, 0.5]
    dfs = [30_000, 9_000, 14_000, 4_000, 600, 50, 1_200, 5, 300, 2_500]
    nd = 4
    dl, offs, docs, fields, tfs = _fields_corpus(O, n_docs, n_fields, dfs, 12)
    sh = S.Shard(0)
    e = int(offs[nd])
    sh.upload_lexical_fields(n_docs, dl, boost, offs[:nd + 1], docs[:e], fields[:e], tfs[:e])
    assert sh.fields_info()[1]  # merged lists
    mid = nd + 3  # two appends: the ids continue
    first = sh.append_sparse_fields(offs[nd:mid + 1] - offs[nd], docs[e:int(offs[mid])], fields[e:int(offs[mid])], tfs[e:int(offs[mid])])
    second = sh.append_sparse_fields(offs[mid:] - offs[mid], docs[int(offs[mid]):], fields[int(offs[mid]):], tfs[int(offs[mid]):])
    assert (first, second) == (nd, mid) and sh.sparse_info()[0] == len(dfs) - nd
    assert [int(x) for x in sh.posting_count(np.arange(len(dfs)))] == dfs  # docs holding the term in any field
    cases = [([0, 4], []), ([6, 1, 2], []), ([9, 6], []), ([4], []), ([7, 5, 0], []), ([3, 9], [1]), ([2, 1], [9]), ([6, 9, 0], [4, 3]),
             ([0, 1], []), ([8, 9, 6, 4], [])]
    gone = list(range(3, n_docs, 97))
    for deleted in ((), gone):
        sh.set_deleted(deleted)
        for qt, oop in ((S.QueryType.Union, O.OP_OR), (S.QueryType.Intersection, O.OP_AND)):
            for strat in (N.BM25_AUTO, N.BM25_EXHAUSTIVE):
                sh.set_strategy(strat)
                q = sh.make_queries([c[0] for c in cases], qt, [c[1] for c in cases])
                for k in (10, 150):
                    for rt in (S.ResultType.TopkCount, S.ResultType.Count):
                        doc, score, cnt, tot = sh.search_lexical_batch(q, k, rt, reference_shortcuts=False)
                        for i, (pos, neg) in enumerate(cases):
                            od, os_, otot, _ = O.search_fields_exhaustive(n_docs, dl, boost, offs, docs, fields, tfs, pos, oop, k, neg, deleted)
                            assert int(tot[i]) == otot, (pos, neg, qt, strat, rt, k, int(tot[i]), otot)
                            if rt != S.ResultType.Count:
                                _check_topk(doc[i], score[i], cnt[i], od, os_)
    sh.set_strategy(N.BM25_AUTO)
    sh.set_deleted(())
    # field filters (intersections and single terms: every term must stand in a listed field -- a sparse posting records its fields)
    fcases = [([0, 4], []), ([6, 9], []), ([4], []), ([9, 6, 1], []), ([6, 0], [9]), ([2, 1], [6]), ([9], [0]), ([0, 1], [])]
    for deleted in ((), gone):
        sh.set_deleted(deleted)
        for filt in ((0,), (1, 2), (2,)):
            q = sh.make_queries([c[0] for c in fcases], S.QueryType.Intersection, [c[1] for c in fcases], field_filter=filt)
            for rt in (S.ResultType.TopkCount, S.ResultType.Count):
                doc, score, cnt, tot = sh.search_lexical_batch(q, 10, rt, reference_shortcuts=False)
                for i, (pos, neg) in enumerate(fcases):
                    od, os_, otot, _ = O.search_fields_exhaustive(n_docs, dl, boost, offs, docs, fields, tfs, pos, O.OP_AND, 10, neg, deleted,
                                                                  field_filter=filt)
                    assert int(tot[i]) == otot, (pos, neg, filt, rt, int(tot[i]), otot)
                    if rt != S.ResultType.Count:
                        _check_topk(doc[i], score[i], cnt[i], od, os_)
    # a UNION of several terms under a filter that names a sparse term: the reference's own sub-queries (every subset of the terms as
    # a filtered intersection, a doc keeps its best) behind the ABI -- the sum over the doc's terms that stand in a listed field, a doc
    # none of whose terms passes is no result; totals: two terms |pass(X) u pass(Y)|, more the unfiltered union
    ucases = [([0, 4], []), ([6, 9], []), ([0, 1, 4], []), ([9, 6, 1, 0], []), ([4, 0], [1]), ([6, 2], [9]), ([0, 1], []), ([8, 4, 6, 9, 2], [])]
    for deleted in ((), gone):
        sh.set_deleted(deleted)
        gone_set = set(deleted)
        per_term = {}
        for t in range(len(dfs)):
            d, s_, _, _ = O.search_fields_exhaustive(n_docs, dl, boost, offs, docs, fields, tfs, [t], O.OP_OR, n_docs, (), deleted)
            a, b = int(offs[t]), int(offs[t + 1])
            per_term[t] = (dict(zip(d.tolist(), s_.tolist())), docs[a:b], fields[a:b])
        for filt in ((0,), (1, 2)):
            q = sh.make_queries([c[0] for c in ucases], S.QueryType.Union, [c[1] for c in ucases], field_filter=filt)
            for k in (10, 40):
                for rt in (S.ResultType.TopkCount, S.ResultType.Count, S.ResultType.Topk):
                    doc, score, cnt, tot = sh.search_lexical_batch(q, k, rt, reference_shortcuts=False)
                    for i, (pos, neg) in enumerate(ucases):
                        sc, passing, present = {}, [], set()
                        for t in pos:
                            ts, dd, ff = per_term[t]
                            pas = set(dd[np.isin(ff, list(filt))].tolist()) - gone_set
                            passing.append(pas)
                            present |= set(dd.tolist()) - gone_set
                            for d in pas:
                                sc[d] = np.float32(sc.get(d, np.float32(0)) + np.float32(ts[d]))
                        banned = set()
                        for t in neg:
                            banned |= set(per_term[t][1].tolist())
                        want = sorted(((d, float(v)) for d, v in sc.items() if d not in banned), key=lambda e: (-e[1], e[0]))[:k]
                        if rt != S.ResultType.Topk:
                            exp = len((passing[0] | passing[1]) - banned) if len(pos) == 2 else len(present - banned)
                            assert int(tot[i]) == exp, (pos, neg, filt, rt, k, int(tot[i]), exp)
                        if rt != S.ResultType.Count:
                            n = int(cnt[i])
                            assert n == len(want), (pos, neg, filt, k, n, len(want))
                            assert np.allclose(score[i, :n], [w[1] for w in want], rtol=1e-4), (pos, neg, filt, k)
                            kth = want[-1][1] if want else 0.0
                            band = abs(kth) * 2e-4
                            assert {int(d) for d, v in zip(doc[i, :n], score[i, :n]) if v > kth + band} == {d for d, v in want if v > kth + band}
    sh.set_deleted(())
    one = sh.search_lexical_shard([0, 4], S.QueryType.Union, 0, 10, S.ResultType.TopkCount, strict=True, field_filter=[1, 2])  # a call of its own
    q = sh.make_queries([[0, 4]], S.QueryType.Union, field_filter=(1, 2))
    doc, score, cnt, tot = sh.search_lexical_batch(q, 10, S.ResultType.TopkCount, reference_shortcuts=False)
    assert [r.doc_id for r in one.results] == doc[0, :int(cnt[0])].tolist() and one.result_count_total == int(tot[0]) and int(cnt[0]) > 0
    # 6 .. 10 terms (63 .. 1023 sub-queries in one batch): composed behind the ABI since round 6 -- the range of union_docid_3
    from test_gpu_shape_sweep import _gated_union_oracle, _check
    per_term = {}
    for t in range(len(dfs)):
        d, s_, _, _ = O.search_fields_exhaustive(n_docs, dl, boost, offs, docs, fields, tfs, [t], O.OP_OR, n_docs, (), ())
        a, b = int(offs[t]), int(offs[t + 1])
        per_term[t] = (dict(zip(d.tolist(), s_.tolist())), docs[a:b], fields[a:b])
    for terms, neg in (([0, 4, 1, 2, 3, 5], []), (list(range(10)), []), ([9, 8, 7, 6, 5, 4, 3], [0])):
        doc, score, cnt, tot = sh.search_lexical_batch(sh.make_queries([terms], S.QueryType.Union, [neg], field_filter=[0]), 10, reference_shortcuts=False)
        od, os_, otot = _gated_union_oracle(per_term, terms, neg, (0,), set(), 10)
        _check(doc[0], score[0], cnt[0], tot[0], od, os_, otot, S.ResultType.TopkCount, 10, S, ("composed", terms, neg))
    sh.close()
