"""Round 4 (-m gpu): NOT lists, tombstones and exact counts on the 16-bit tile of the exhaustive strategy
(bm25_scan16.hip: candidate-path exclusions, EXCL instances) -- against the oracle and, bit for bit, against the f32-tile
kernel they replace (SS_BM25_EXHAUSTIVE_F32)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-4
VOC = [0, 1500, 2500, 3000, 3300, 3600, 3800, 3900, 4000, 4050, 4095]  # df from 0.05 % to 20 %


@pytest.fixture(scope="module")
def S():
    import seekstorm_amd
    return seekstorm_amd


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


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


def _oracle_check(S, O, osh, cs, oop, rt, got):
    doc, score, cnt, tot = got
    for i, (pos, neg) in enumerate(cs):
        od, os_, otot = osh.search_exhaustive(pos, oop, 10, not_terms=neg)
        if rt != S.ResultType.Topk:
            assert int(tot[i]) == otot, (pos, neg, rt)
        if rt != S.ResultType.Count:
            _check_topk(doc[i], score[i], cnt[i], od, os_)


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
                    sh.set_strategy(N.BM25_EXHAUSTIVE)
                    a = sh.search_lexical_batch(q, 10, rt)
                    sh.set_strategy(N.BM25_EXHAUSTIVE_F32)
                    b = sh.search_lexical_batch(q, 10, rt)
                    _same(a, b, (qt, rt, bool(gone)))
                    _oracle_check(S, O, osh, cs, oop, rt, a)
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


def test_drop_in_rehearsal_on_a_million_doc_index_bin(S, O):
    """index.bin (1 M docs, >= 1 M keys, clustered doc ids, NgramFF | NgramFFF keys, positions) + vector.bin +
    delete.bin as the reference lays them out -> ss_index_bin_open -> tier -> upload with positions; 2-term ANDs, 3-term ORs (rare
    terms from the sparse tier included), phrases over n-gram keys, vector and hybrid queries, every answer against the oracle;
    64 concurrent callers through Index::search of the C++ mirror.  tools/real_format.py holds the rehearsal (bench.py runs it too)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import real_format
    r = real_format.run(n_docs=1_000_000, vocab=1_000_000, n_queries=64, seconds=0.6)
    assert r["files"]["keys"] >= 1_000_000 and r["files"]["ngram_keys"] > 10_000
    assert r["open"]["sparse_terms"] > 900_000 and r["open"]["dense_terms"] > 1000
    assert r["queries"]["phrases_with_ngram_keys"] > 0 and r["queries"]["ors_naming_a_sparse_term"] > 0
    assert set(r["parity"]["queries"]) == {"and2", "or3", "phrase", "vector", "hybrid"}
    assert all(v["errors"] == 0 for v in r["concurrent_callers"].values())
