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


def test_append_level_by_level_equals_the_one_shot_upload(S, O):
    """ss_bm25_append_level (commit.rs:142-148): the image after every commit answers exactly like a one-shot upload of the docs
    committed so far -- ids, scores (==), counts, both strategies; the vocabulary grows on the way; the last, partial level is
    re-committed with more docs; NOT terms / tombstones ride along; the device rebuild takes milliseconds"""
    from seekstorm_amd import _native as N
    n_docs = 300_000
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = O.lex_corpus(n_docs, VOC)
    levels = _level_slices(n_docs, offs, docs, tfs)
    inc, ref = S.Shard(0), S.Shard(0)
    tl_or = [[10, 9, 8], [7, 3], [10], [9, 8, 7, 6], [5, 4, 3, 2, 1], [10, 2]]
    tl_and = [[10, 9], [10, 9, 8], [8, 5]]

    committed = []  # the levels as they were handed over: (offs, docs, tfs) -- the one-shot image holds exactly these postings

    def same(n_now, nt_now, what):
        so = np.zeros(nt_now + 1, np.uint64)
        dd, tt = [], []
        for t in range(nt_now):
            for lo_, do_, to_ in committed:
                if t < len(lo_) - 1:
                    dd.append(do_[int(lo_[t]):int(lo_[t + 1])]); tt.append(to_[int(lo_[t]):int(lo_[t + 1])])
            so[t + 1] = sum(len(x) for x in dd)
        ref.upload_lexical(n_now, dl[:n_now], so, np.concatenate(dd), np.concatenate(tt))
        ok_terms = lambda tl: [q for q in tl if max(q) < nt_now]
        for qt, tls in ((S.QueryType.Union, ok_terms(tl_or)), (S.QueryType.Intersection, ok_terms(tl_and))):
            if not tls:
                continue
            for strat in (N.BM25_AUTO, N.BM25_EXHAUSTIVE):
                inc.set_strategy(strat); ref.set_strategy(strat)
                for rt in (S.ResultType.TopkCount, S.ResultType.Topk):
                    x = inc.search_lexical_batch(inc.make_queries(tls, qt), 10, rt)
                    y = ref.search_lexical_batch(ref.make_queries(tls, qt), 10, rt)
                    for u, v, name in zip(x, y, ("doc", "score", "count", "total")):
                        assert np.array_equal(u, v), (what, qt, strat, rt, name)
    try:
        nt_first = 7  # the first two commits know 7 terms, the vocabulary then grows to 11
        for lv, (lo, hi, lo_, do_, to_) in enumerate(levels):
            nt_now = nt_first if lv < 2 else len(VOC)
            if nt_now < len(VOC):
                cut = int(lo_[nt_now])
                lo_, do_, to_ = lo_[:nt_now + 1], do_[:cut], to_[:cut]
            if lv == len(levels) - 1:  # the last level: first committed half full, then re-committed whole
                half = lo + (hi - lo) // 2
                keep = do_ < half
                lo_h = np.zeros(len(lo_), np.uint64)
                for t in range(len(lo_) - 1):
                    lo_h[t + 1] = lo_h[t] + int(keep[int(lo_[t]):int(lo_[t + 1])].sum())
                inc.append_level(lv, dl[lo:half], lo_h, do_[keep], to_[keep])
                committed.append((lo_h, do_[keep], to_[keep]))
                same(half, nt_now, ("partial", lv))
                committed.pop()
            inc.append_level(lv, dl[lo:hi], lo_, do_, to_)
            committed.append((lo_, do_, to_))
            nl, raw_b, ms_all, ms_dev = inc.incremental_info()
            assert nl == lv + 1 and raw_b > 0 and 0 < ms_dev <= ms_all < 2000
            same(hi, nt_now, ("level", lv))
        # tombstones set before a commit survive it; NOT terms work on the rebuilt image
        gone = list(range(5, n_docs, 211))
        inc.set_deleted(gone); ref.set_deleted(gone)
        lo, hi, lo_, do_, to_ = levels[-1]
        inc.append_level(len(levels) - 1, dl[lo:hi], lo_, do_, to_)  # re-commit once more
        q = ([[10, 9, 8], [9, 7]], S.QueryType.Union, [[7], [10]])
        x = inc.search_lexical_batch(inc.make_queries(*q), 10)
        y = ref.search_lexical_batch(ref.make_queries(*q), 10)
        assert all(np.array_equal(u, v) for u, v in zip(x, y))
        # what the ABI refuses: a gap, a shrinking vocabulary, docs outside the level, an image from another builder
        with pytest.raises(S.SeekStormHipError):
            inc.append_level(len(levels) + 1, dl[:10], np.zeros(len(VOC) + 1, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.uint16))
        with pytest.raises(S.SeekStormHipError):
            inc.append_level(len(levels) - 1, dl[lo:hi], lo_[:4], do_[:int(lo_[3])], to_[:int(lo_[3])])
        with pytest.raises(S.SeekStormHipError):
            inc.append_level(len(levels) - 1, dl[lo:hi], lo_, do_ - np.uint32(70000), to_)
        with pytest.raises(S.SeekStormHipError):
            ref.append_level(0, dl[:100], np.zeros(len(VOC) + 1, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.uint16))
    finally:
        inc.close()
        ref.close()


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
