"""The exhaustive strategy's 16-bit tile (bm25_scan16_kernel): exact counts, intersections, NOT lists and tombstones inside the streaming loop, against the f32 tile and the oracle; the clustered corpus generator."""
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


def test_exhaustive_16bit_scan_counts_and_intersections(S, O):
    """round 3: the 16-bit scan also serves exact union counts (first-touch counting) and intersections of 2 / 3 terms (entries with
    a level).  Lists from 0.3 % to 60 % of the docs -- sparse segments, and segments longer than the register chunks (> 12.5 % /
    18.75 % of a sub-block: the synchronously streamed remainder) -- against the oracle's exhaustive answers: exact counts, bit-exact
    id sets where the intersection is smaller than k, scores 1e-4; Topk / TopkCount / Count, k = 10 and 64; the pruned strategy
    returns the same lists bit for bit; with tombstones the counts fall back to the f32 kernel and stay exact"""
    from seekstorm_amd import _native as N
    n_docs = 150_000
    dfs = [0.003, 0.01, 0.03, 0.08, 0.15, 0.22, 0.35, 0.6, 0.5, 0.12]
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = _dense_corpus(O, n_docs, dfs)
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, offs, docs, tfs)
    osh = O.Shard(n_docs, dl, offs, docs, tfs)
    rng = np.random.default_rng(3)
    nt_all = len(dfs)
    for nterms, op, qt in ((2, O.OP_AND, S.QueryType.Intersection), (3, O.OP_AND, S.QueryType.Intersection),
                           (2, O.OP_OR, S.QueryType.Union), (3, O.OP_OR, S.QueryType.Union), (4, O.OP_OR, S.QueryType.Union)):
        tl = [[int(x) for x in rng.choice(nt_all, nterms, replace=False)] for _ in range(40)]
        q = sh.make_queries(tl, qt)
        for k in (10, 64, 100, 128):
            want = [osh.search_exhaustive(t, op, k) for t in tl]
            sh.set_strategy(N.BM25_AUTO)
            pd, ps, pc, pt = sh.search_lexical_batch(q, k, S.ResultType.TopkCount, reference_shortcuts=False)
            sh.set_strategy(N.BM25_EXHAUSTIVE)
            for rt in (S.ResultType.TopkCount, S.ResultType.Topk, S.ResultType.Count):
                d, s_, c, t = sh.search_lexical_batch(q, k, rt, reference_shortcuts=False)
                for i in range(len(tl)):
                    od, os_, otot = want[i]
                    if rt != S.ResultType.Topk:
                        assert int(t[i]) == otot, (nterms, op, k, int(rt), i, int(t[i]), otot)
                    if rt != S.ResultType.Count:
                        assert c[i] == len(od)
                        assert np.allclose(s_[i, :c[i]], os_, rtol=1e-4)
                        if len(od) < k:
                            assert set(d[i, :c[i]].tolist()) == set(int(x) for x in od)
                if rt == S.ResultType.TopkCount:
                    assert np.array_equal(d, pd) and np.array_equal(s_, ps) and np.array_equal(t, pt)  # both strategies, bit for bit
    # tombstones: a deleted doc neither counts nor ranks -- the 16-bit scan's count mode steps aside, the answers stay exact
    gone = [int(x) for x in rng.choice(n_docs, 5000, replace=False)]
    sh.set_deleted(gone)
    osh.set_deleted(gone)
    tl = [[7, 8], [4, 5, 6], [1, 9]]
    for op, qt in ((O.OP_OR, S.QueryType.Union), (O.OP_AND, S.QueryType.Intersection)):
        q = sh.make_queries([t for t in tl if len(t) == (2 if op == O.OP_AND else len(t))], qt)
        tls = [t for t in tl if len(t) == (2 if op == O.OP_AND else len(t))]
        d, s_, c, t = sh.search_lexical_batch(q, 10, S.ResultType.TopkCount, reference_shortcuts=False)
        for i, terms in enumerate(tls):
            od, os_, otot = osh.search_exhaustive(terms, op, 10)
            assert int(t[i]) == otot and np.allclose(s_[i, :c[i]], os_, rtol=1e-4)
    sh.close()


def test_exclusions_on_the_16_bit_tile_equal_the_f32_tile_and_the_oracle(S, O, lex):
    """unions of <= 4 lists (+ 5 / 6 for top-k) and intersections of 2 / 3 with NOT lists and / or tombstones, every result type:
    EXHAUSTIVE (16-bit tile: exclusions in the candidate path, EXCL count instances) == EXHAUSTIVE_F32 == oracle"""
    from seekstorm_amd import _native as N
    sh, osh, n_docs = lex
    unions = [([10, 9, 8], [7]), ([10, 9], [8]), ([10], [9]), ([9, 8, 7, 6], [10]), ([10, 9, 8], []), ([6, 5], [10]), ([10, 9, 8], [3])]
    unions_2not = [([10, 9, 8], [7, 6]), ([10, 9], [8, 2]), ([9], [10, 1]), ([10, 9, 8], [])]
    wide = [([10, 9, 8, 7, 6], [5]), ([10, 9, 8, 7, 6, 5], [4, 3]), ([5, 4, 3, 2, 1], [10])]
    ands2 = [([10, 9], [8]), ([10, 8], [7]), ([9, 7], [10]), ([10, 9], [])]
    ands3 = [([10, 9, 8], [7]), ([10, 9, 7], [8]), ([10, 9, 8], [])]
    rng = np.random.default_rng(5)
    gone_sets = [[], sorted(set(int(x) for x in rng.choice(n_docs, size=n_docs // 50, replace=False)) | set(range(0, 4096, 3)))]
    try:
        for gone in gone_sets:
            sh.set_deleted(gone)
            osh.set_deleted(gone)
            for cs, qt, oop, rts in ((unions, S.QueryType.Union, O.OP_OR, (S.ResultType.Topk, S.ResultType.TopkCount, S.ResultType.Count)),
                                     (unions_2not, S.QueryType.Union, O.OP_OR, (S.ResultType.Topk, S.ResultType.TopkCount)),
                                     (wide, S.QueryType.Union, O.OP_OR, (S.ResultType.Topk,)),
                                     (ands2, S.QueryType.Intersection, O.OP_AND, (S.ResultType.Topk, S.ResultType.TopkCount, S.ResultType.Count)),
                                     (ands3, S.QueryType.Intersection, O.OP_AND, (S.ResultType.Topk, S.ResultType.TopkCount))):
                q = sh.make_queries([c[0] for c in cs], qt, [c[1] for c in cs])
                for rt in rts:
                    for k in (10, 100):  # (k = 100: two keys per lane in the candidate path, no k-lane cut)
                        sh.set_strategy(N.BM25_EXHAUSTIVE)
                        a = sh.search_lexical_batch(q, k, rt)
                        sh.set_strategy(N.BM25_EXHAUSTIVE_F32)
                        b = sh.search_lexical_batch(q, k, rt)
                        _same(a, b, (qt, rt, bool(gone), k))
                        _oracle_check(S, O, osh, cs, oop, rt, a, k)
    finally:
        sh.set_strategy(0)
        sh.set_deleted([])
        osh.set_deleted([])


def test_the_best_docs_of_a_sub_block_are_tombstoned(S, O):
    """the candidate path's k-lane cut must not be raised by docs that cannot be results.  The 40 best docs of a query
    (all inside two sub-blocks) are deleted, then the 40 best of what is left, ...: every answer equals the oracle's."""
    from seekstorm_amd import _native as N
    n_docs = 8192
    dl = O.lex_doclen(n_docs)
    voc = [3900, 4000, 4095]
    offs, docs, tfs = O.lex_corpus(n_docs, voc)
    osh = O.Shard(n_docs, dl, offs, docs, tfs)
    sh = S.Shard(0)
    try:
        sh.upload_lexical(n_docs, dl, offs, docs, tfs)
        for qt, oop, terms in ((S.QueryType.Union, O.OP_OR, [0, 1, 2]), (S.QueryType.Union, O.OP_OR, [2]), (S.QueryType.Intersection, O.OP_AND, [1, 2])):
            gone = []
            sh.set_deleted(gone)
            osh.set_deleted(gone)
            q = sh.make_queries([terms], qt)
            for _ in range(6):
                od, os_, _ = osh.search_exhaustive(terms, oop, 40)
                gone = sorted(set(gone) | {int(d) for d in od})
                sh.set_deleted(gone)
                osh.set_deleted(gone)
                for strat in (N.BM25_EXHAUSTIVE, N.BM25_AUTO):
                    sh.set_strategy(strat)
                    for rt in (S.ResultType.Topk, S.ResultType.TopkCount):
                        doc, score, cnt, tot = sh.search_lexical_batch(q, 10, rt)
                        od10, os10, otot = osh.search_exhaustive(terms, oop, 10)
                        assert not set(map(int, doc[0][:cnt[0]])) & set(gone)
                        _check_topk(doc[0], score[0], cnt[0], od10, os10)
                        if rt == S.ResultType.TopkCount:
                            assert int(tot[0]) == otot
    finally:
        sh.close()


def test_dense_not_list_and_heavy_tombstones(S, O, lex):
    """a NOT list denser than its register chunks (20 % of the docs: the synchronous remainder) and a shard with a third of its docs
    deleted, counts included"""
    from seekstorm_amd import _native as N
    sh, osh, n_docs = lex
    cs = [([8, 7, 6], [10]), ([9, 5], [10]), ([3, 2], [10]), ([9, 8, 7, 6], [10])]
    gone = list(range(1, n_docs, 3))
    try:
        sh.set_deleted(gone)
        osh.set_deleted(gone)
        q = sh.make_queries([c[0] for c in cs], S.QueryType.Union, [c[1] for c in cs])
        for rt in (S.ResultType.Topk, S.ResultType.TopkCount, S.ResultType.Count):
            sh.set_strategy(N.BM25_EXHAUSTIVE)
            a = sh.search_lexical_batch(q, 10, rt)
            sh.set_strategy(N.BM25_EXHAUSTIVE_F32)
            _same(a, sh.search_lexical_batch(q, 10, rt), rt)
            _oracle_check(S, O, osh, cs, O.OP_OR, rt, a)
    finally:
        sh.set_strategy(0)
        sh.set_deleted([])
        osh.set_deleted([])


def test_clustered_generator_device_equals_oracle(S, O):
    """seeds with bit 63 set: a term's density varies with the doc's cluster (device lex_cluster_thresh == oracle so_lex_cluster_thresh);
    the corpus really is clustered (per-block posting counts differ many-fold), and the strategies agree on it (block maxima in use)"""
    from seekstorm_amd import _native as N
    n_docs, nt = 300_000, 16
    th = O.term_thresholds(nt)
    seed = O.LEX_SEED_CLUSTERED
    a, b = S.Shard(0), S.Shard(0)
    try:
        a.synth_lexical(seed, n_docs, th, O.len_table())
        dl = O.lex_doclen(n_docs, seed)
        offs, docs, tfs = O.lex_corpus(n_docs, list(range(nt)), seed=seed, thresholds=th)
        b.upload_lexical(n_docs, dl, offs, docs, tfs)
        assert a.lexical_info() == b.lexical_info()
        assert np.array_equal(a.posting_count(np.arange(nt)), b.posting_count(np.arange(nt)))
        d15 = docs[int(offs[15]):int(offs[16])]
        per_block = np.bincount(d15 >> 13, minlength=n_docs >> 13)[:n_docs >> 13]
        assert per_block.max() > 8 * max(per_block.min(), 1)
        osh = O.Shard(n_docs, dl, offs, docs, tfs)
        tl = [[15, 14, 9], [13, 2], [15], [12, 11, 10, 3], [15, 14]]
        for qt, oop in ((S.QueryType.Union, O.OP_OR), (S.QueryType.Intersection, O.OP_AND)):
            res = {}
            for strat in (N.BM25_AUTO, N.BM25_EXHAUSTIVE, N.BM25_PRUNED):
                a.set_strategy(strat); b.set_strategy(strat)
                ra = a.search_lexical_batch(a.make_queries(tl, qt), 10)
                rb = b.search_lexical_batch(b.make_queries(tl, qt), 10)
                assert all(np.array_equal(x, y) for x, y in zip(ra, rb))
                res[strat] = ra
            for strat in (N.BM25_EXHAUSTIVE, N.BM25_PRUNED):
                assert all(np.array_equal(x, y) for x, y in zip(res[N.BM25_AUTO], res[strat]))
            for i, q in enumerate(tl):
                od, os_, otot = osh.search_exhaustive(q, oop, 10)
                assert int(res[N.BM25_AUTO][3][i]) == otot
                _check_topk(res[N.BM25_AUTO][0][i], res[N.BM25_AUTO][1][i], res[N.BM25_AUTO][2][i], od, os_)
    finally:
        a.close(); b.close()
