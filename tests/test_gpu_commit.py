"""Incremental commits (ss_bm25_append_level[_positions], ss_bm25_append_sparse_level, ss_vec_append_rows): an image grown level by level answers like the one-shot upload of the same docs."""
import ctypes as C
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


@pytest.mark.parametrize("euclid", [False, True])
def test_vector_rows_appended_level_by_level_equal_the_one_shot_upload(S, O, euclid):
    """ss_vec_append_rows: levels of vector records written behind the image in place (f32; doc ids with several records per doc,
    field ids, a cluster structure for the ANN modes) -- the answers are those of the one-shot upload, bit for bit, after every level"""
    dim = 96
    sizes = [5000, 65536, 700, 129, 20000]
    rng = np.random.default_rng(12)
    n_all = sum(sizes)
    rows = O.vec_gen(O.VEC_SEED, 0, n_all, dim)
    ids = np.concatenate([np.sort(rng.integers(0, max(n // 2, 1), n)).astype(np.uint32) + np.uint32(l << 16) for l, n in enumerate(sizes)])
    fld = rng.integers(0, 3, n_all).astype(np.uint16)
    child = []
    for n in sizes:  # clusters of a level: a few, uneven
        cuts = np.sort(rng.choice(np.arange(1, n), size=min(6, n - 1), replace=False))
        child.append(np.diff(np.concatenate([[0], cuts, [n]])).astype(np.uint32))
    qs = O.vec_gen(O.VECQ_SEED, 0, 9, dim)
    inc = S.Shard(0)
    if euclid:
        inc.set_vector_similarity("euclidean")
    inc.upload_vectors(rows[:sizes[0]], ids[:sizes[0]])
    inc.set_fields(fld[:sizes[0]])
    inc.set_clusters([len(child[0])], child[0])
    at = sizes[0]
    for l in range(1, len(sizes)):
        n = sizes[l]
        inc.append_vector_rows(rows[at:at + n], row_doc_ids=ids[at:at + n], row_field=fld[at:at + n], child_count=child[l])
        at += n
        ref = S.Shard(0)
        if euclid:
            ref.set_vector_similarity("euclidean")
        ref.upload_vectors(rows[:at], ids[:at])
        ref.set_fields(fld[:at])
        ref.set_clusters([len(c) for c in child[:l + 1]], np.concatenate(child[:l + 1]))
        assert inc.cluster_info() == ref.cluster_info() and inc.vector_count == at
        for k in (10, 100):
            _same_answers(inc.search_vector_batch(qs, k), ref.search_vector_batch(qs, k))
            _same_answers(inc.search_vector_batch(qs, k, field_filter=[1]), ref.search_vector_batch(qs, k, field_filter=[1]))
            if not euclid:
                _same_answers(inc.search_vector_batch(qs, k, ann_mode=S.AnnMode(n_probe=2), with_clusters=True),
                              ref.search_vector_batch(qs, k, ann_mode=S.AnnMode(n_probe=2), with_clusters=True))
        ref.close()
    # the level must bring what the image carries per row
    from seekstorm_amd import _native as N
    with pytest.raises(N.SeekStormHipError):
        inc.append_vector_rows(rows[:10], row_doc_ids=ids[:10], row_field=fld[:10])          # no clusters
    with pytest.raises(N.SeekStormHipError):
        inc.append_vector_rows(rows[:10], row_field=fld[:10], child_count=[10])               # no doc ids
    inc.close()


def test_i8_vector_rows_appended_equal_the_one_shot_upload(S, O):
    """... for Precision::I8 records with per-record scales (the fragment-ordered image: rows scattered into their tiles), plain ids"""
    dim, sizes = 200, [3000, 65536, 130, 9000]
    rng = np.random.default_rng(5)
    n_all = sum(sizes)
    rows = rng.integers(-127, 128, (n_all, dim)).astype(np.int8)
    scale = (rng.random(n_all) * 0.01 + 0.001).astype(np.float32)
    qs = rng.integers(-127, 128, (7, dim)).astype(np.int8)
    qsc = (rng.random(7) * 0.01 + 0.001).astype(np.float32)
    for euclid in (False, True):
        inc = S.Shard(0)
        if euclid:
            inc.set_vector_similarity("euclidean")
        inc.upload_vectors_i8(rows[:sizes[0]], None if euclid else scale[:sizes[0]])
        inc.reserve_vector_rows(70_000)  # the first two appends find room, the third grows the image
        at = sizes[0]
        for n in sizes[1:]:
            inc.append_vector_rows(rows[at:at + n], row_scale=None if euclid else scale[at:at + n])
            at += n
            ref = S.Shard(0)
            if euclid:
                ref.set_vector_similarity("euclidean")
            ref.upload_vectors_i8(rows[:at], None if euclid else scale[:at])
            for k in (10, 100):
                _same_answers(inc.search_vector_batch_i8(qs, k, None if euclid else qsc), ref.search_vector_batch_i8(qs, k, None if euclid else qsc))
            ref.close()
        inc.close()


def test_append_levels_with_positions_answer_phrases_like_the_one_shot_upload(S, O):
    """ss_bm25_append_level_positions: an image that grows by commits serves PHRASE queries -- the position arrays (pool, per-slot end
    offsets, per-term bases) are rebuilt on the device from the levels' own pools.  After every level: phrases (2 - 4 words, repeated
    words, NOT terms), unions and intersections == a one-shot upload with positions of the docs committed so far"""
    from test_gpu_phrase import _corpus
    n_docs = 200_000  # 3 full levels + a partial one
    dfs = [30_000, 22_000, 40_000, 9_000, 700]
    plant = [([0, 1], 500), ([0, 1, 2], 200), ([2, 0, 2], 150), ([3, 4], 80), ([1, 1], 90), ([4, 0, 1, 3], 40)]
    dl, offs, docs, tfs, positions = _corpus(O, n_docs, dfs, 41, plant)
    pstart = np.zeros(len(docs) + 1, np.int64)
    pstart[1:] = np.cumsum(tfs.astype(np.int64))
    nt = len(dfs)
    inc = S.Shard(0)
    phrases = [[0, 1], [1, 0], [0, 1, 2], [2, 0, 2], [3, 4], [1, 1], [4, 0, 1, 3], [2, 2]]
    sets = [[0, 1], [2, 3, 4], [0, 4]]
    n_levels = (n_docs + 65535) >> 16
    for l in range(n_levels):
        d0, d1 = l << 16, min(n_docs, (l + 1) << 16)
        lo, ld, lt, lp = [0], [], [], []
        for t in range(nt):
            a, b = int(offs[t]), int(offs[t + 1])
            i0, i1 = a + int(np.searchsorted(docs[a:b], d0)), a + int(np.searchsorted(docs[a:b], d1))
            ld.append(docs[i0:i1]); lt.append(tfs[i0:i1]); lp.append(positions[pstart[i0]:pstart[i1]]); lo.append(lo[-1] + (i1 - i0))
        inc.append_level(l, dl[d0:d1], np.asarray(lo, np.uint64), np.concatenate(ld), np.concatenate(lt), positions=np.concatenate(lp))
        # the reference image: everything committed so far, uploaded at once
        ro, rd, rt_, rp = [0], [], [], []
        for t in range(nt):
            a, b = int(offs[t]), int(offs[t + 1])
            i1 = a + int(np.searchsorted(docs[a:b], d1))
            rd.append(docs[a:i1]); rt_.append(tfs[a:i1]); rp.append(positions[pstart[a]:pstart[i1]]); ro.append(ro[-1] + (i1 - a))
        ref = S.Shard(0)
        ref.upload_lexical(d1, dl[:d1], np.asarray(ro, np.uint64), np.concatenate(rd), np.concatenate(rt_), np.concatenate(rp))
        assert inc.fields_info()[2] == 1  # positions present
        for k in (10, 100):
            for rt in (S.ResultType.TopkCount, S.ResultType.Count):
                a = inc.search_lexical_batch(inc.make_queries(phrases, S.QueryType.Phrase), k, rt)
                b = ref.search_lexical_batch(ref.make_queries(phrases, S.QueryType.Phrase), k, rt)
                _same(a, b, ("phrases", l, k, rt))
        a = inc.search_lexical_batch(inc.make_queries([[0, 1], [3, 4]], S.QueryType.Phrase, [[2], [0]]), 10)
        b = ref.search_lexical_batch(ref.make_queries([[0, 1], [3, 4]], S.QueryType.Phrase, [[2], [0]]), 10)
        _same(a, b, ("phrases with NOT terms", l))
        for qt in (S.QueryType.Union, S.QueryType.Intersection):
            _same(inc.search_lexical_batch(inc.make_queries(sets, qt), 10), ref.search_lexical_batch(ref.make_queries(sets, qt), 10), ("sets", l, qt))
        ref.close()
    assert int(inc.search_lexical_batch(inc.make_queries([[0, 1]], S.QueryType.Phrase), 10)[3][0]) >= 500
    # a level without positions after levels with them is refused
    from seekstorm_amd import _native as N
    with pytest.raises(N.SeekStormHipError):
        inc.append_level(n_levels - 1, dl[(n_levels - 1) << 16:], np.asarray(lo, np.uint64), np.concatenate(ld), np.concatenate(lt))
    inc.close()


def test_append_levels_with_ngram_key_positions_against_the_oracle(S, O):
    """the incremental image over a DEFAULT-index vocabulary: n-gram keys as their component terms, the key's own positions behind the
    first component (npos), committed level by level with positions -- phrases resolved the way the query tokenizer resolves them
    (greedy trigram / bigram keys over frequent words) against the oracle's phrase check over the mini indexer's own lists"""
    from oracle import textindex as TI
    from seekstorm_amd.search import idf_f32
    T = TI.TextCorpus(5, 140_000, 3000, n_frequent=12, mean_len=8.0, topic_share=0.4)
    n_docs = T.n_docs
    term_of, lists = {}, []
    for k in range(T.n_keys):
        if T.key_df(k) == 0:
            continue
        nc = 1 if k < T.vocab else (2 if (T.key_hash(k) & 7) == 1 else 3)
        for c in range(nc):
            docs, tfs, cnt, pos = T.key_postings(k, c, positions=(c == 0))
            term_of[(k, c)] = len(lists)
            lists.append((docs, tfs, cnt if c == 0 else np.zeros(len(docs), np.uint16), pos if c == 0 else np.zeros(0, np.uint16)))
    nt = len(lists)
    pst = []
    for docs, tfs, cnt, pos in lists:
        p = np.zeros(len(docs) + 1, np.int64)
        p[1:] = np.cumsum(cnt.astype(np.int64))
        pst.append(p)
    inc = S.Shard(0)
    n_levels = (n_docs + 65535) >> 16
    for l in range(n_levels):
        d0, d1 = l << 16, min(n_docs, (l + 1) << 16)
        lo, ld, lt, ln, lp = [0], [], [], [], []
        for t, (docs, tfs, cnt, pos) in enumerate(lists):
            i0, i1 = int(np.searchsorted(docs, d0)), int(np.searchsorted(docs, d1))
            ld.append(docs[i0:i1]); lt.append(tfs[i0:i1]); ln.append(cnt[i0:i1]); lp.append(pos[pst[t][i0]:pst[t][i1]]); lo.append(lo[-1] + (i1 - i0))
        inc.append_level(l, T.doclen[d0:d1], np.asarray(lo, np.uint64), np.concatenate(ld), np.concatenate(lt), positions=np.concatenate(lp),
                         npos=np.concatenate(ln))
    offs = np.zeros(nt + 1, np.uint64)
    offs[1:] = np.cumsum([len(x[0]) for x in lists])
    osh = O.Shard(n_docs, T.doclen, offs, np.concatenate([x[0] for x in lists]), np.concatenate([x[1] for x in lists]))
    osh.set_positions(np.concatenate([x[3] for x in lists]), np.concatenate([x[2] for x in lists]))
    rng = np.random.default_rng(2)
    phrases = []
    while len(phrases) < 60:
        d = int(rng.integers(0, n_docs))
        toks = T.doc_tokens(d)
        if len(toks) < 6:
            continue
        st = int(rng.integers(0, len(toks) - 4))
        ents = T.query_entries([int(r) for r in toks[st:st + int(rng.integers(2, 5))]])
        if len(ents) >= 2 and all(e[0] is not None for e in ents):
            phrases.append(ents)
    assert any(len(e[1]) > 1 for q in phrases for e in q)  # some entries are n-gram keys
    idf_of = {}
    qlists = []
    for q in phrases:
        row = []
        for key, ranks in q:
            if len(ranks) == 1:
                row.append(term_of[(key, 0)])
            else:
                comp = tuple(term_of[(key, c)] for c in range(len(ranks)))
                for c, r in enumerate(ranks):
                    idf_of[comp[c]] = float(idf_f32(n_docs, T.key_df(r)))  # idf_ngram_i: from the component TERM's posting count
                row.append(comp)
        qlists.append(row)
    q = inc.make_queries(qlists, S.QueryType.Phrase, idf_of=idf_of)
    doc, score, cnt, tot = inc.search_lexical_batch(q, 10, S.ResultType.TopkCount)
    for i, ents in enumerate(phrases):
        uniq, seq, places, idf, at = [], [], [], [], 0
        for key, ranks in ents:
            ls = [term_of[(key, c)] for c in range(len(ranks))]
            for c, l in enumerate(ls):
                if l not in uniq:
                    uniq.append(l)
                    idf.append(idf_of[l] if l in idf_of else float(idf_f32(n_docs, osh.df(l))))
            seq.append(uniq.index(ls[0])); places.append(at)
            at += len(ranks)
        od, os_, otot = osh.search_phrase_items(uniq, seq, places, 10, idf=idf)
        assert otot >= 1 and int(tot[i]) == otot, (i, ents, int(tot[i]), otot)
        assert cnt[i] == len(od) and np.allclose(score[i][:cnt[i]], os_, rtol=1e-4)
    inc.close()
    T.close()


def test_sparse_tier_committed_level_by_level_equals_the_one_shot_tiered_upload(S, O):
    """ss_bm25_append_sparse_level: an image WITH A SPARSE TIER that grows by commits -- per level the dense terms through append_level,
    the rare terms' postings into their sparse lists (new rare terms join as new lists), the tier's codes re-made on the device as the
    average length moves.  After every level: phrases over both tiers, unions, intersections, NOT terms, k 10 / 100, counts ==
    a one-shot upload of the docs committed so far (dense image + whole sparse lists, positions and all), bit for bit"""
    from test_gpu_phrase import _corpus
    from seekstorm_amd import _native as N
    n_docs, nd = 170_000, 3  # 2 full levels + a partial one; terms 0..2 dense, 3.. sparse
    dfs = [30_000, 22_000, 40_000, 9_000, 700, 60, 3, 2_000]
    plant = [([0, 3], 300), ([3, 4], 80), ([1, 4, 0], 60), ([3, 3], 50), ([7, 0], 100), ([5, 1], 20), ([0, 1], 400)]
    dl, offs, docs, tfs, positions = _corpus(O, n_docs, dfs, 43, plant)
    pstart = np.zeros(len(docs) + 1, np.int64)
    pstart[1:] = np.cumsum(tfs.astype(np.int64))
    lists = []
    for t in range(len(dfs)):
        a, b = int(offs[t]), int(offs[t + 1])
        first = a + int(np.searchsorted(docs[a:b], 65536)) if t == 7 else a  # term 7 enters the vocabulary with level 1
        lists.append((docs[first:b], tfs[first:b], positions[pstart[first]:pstart[b]]))

    def cut(t, d0, d1):
        d, tf, ps = lists[t]
        i0, i1 = int(np.searchsorted(d, d0)), int(np.searchsorted(d, d1))
        pst = np.zeros(len(d) + 1, np.int64)
        pst[1:] = np.cumsum(tf.astype(np.int64))
        return d[i0:i1], tf[i0:i1], ps[pst[i0]:pst[i1]]

    def csr(terms, d0, d1):
        o, dd, tt, pp = [0], [], [], []
        for t in terms:
            d, tf, ps = cut(t, d0, d1)
            dd.append(d); tt.append(tf); pp.append(ps); o.append(o[-1] + len(d))
        return np.asarray(o, np.uint64), np.concatenate(dd), np.concatenate(tt), np.concatenate(pp)

    phrases = [[0, 3], [3, 4], [1, 4, 0], [3, 3], [7, 0], [5, 1], [0, 1], [4, 3]]
    sets = [[0, 3], [4, 1, 2], [5, 6, 0], [3, 4], [7, 3, 1], [6], [7], [1, 2]]
    nots = [([0, 1], [3]), ([3], [0]), ([4, 2], [7]), ([3, 7], [1])]
    inc = S.Shard(0)
    n_levels = (n_docs + 65535) >> 16
    ref = None
    # the last level is committed half full first and re-committed whole (commit.rs:204-206): both calls replace what it brought
    for l, d1 in [(0, 65536), (1, 131072), (2, 150_000), (2, n_docs)]:
        d0 = l << 16
        sparse_terms = list(range(nd, 7 if l == 0 else 8))
        if l == 1:  # the two ABI calls on their own ...
            o, dd, tt, pp = csr(range(nd), d0, d1)
            inc.append_level(l, dl[d0:d1], o, dd, tt, positions=pp)
            o, dd, tt, pp = csr(sparse_terms, d0, d1)
            inc.append_sparse_level(l, o, dd, tt, positions=pp)
        else:       # ... and as the one commit of the mirrors: the level's postings of all known terms in id order
            o, dd, tt, pp = csr(list(range(nd)) + sparse_terms, d0, d1)
            inc.commit_level(l, dl[d0:d1], o, dd, tt, n_dense_terms=nd, positions=pp)
        assert inc.sparse_info()[0] == len(sparse_terms)
        if ref is not None:
            ref.close()
        ref = S.Shard(0)
        o, dd, tt, pp = csr(range(nd), 0, d1)
        ref.upload_lexical(d1, dl[:d1], o, dd, tt, pp)
        o, dd, tt, pp = csr(sparse_terms, 0, d1)
        assert ref.append_sparse(o, dd, tt, positions=pp) == nd
        known = lambda q: all(t < nd + len(sparse_terms) for t in q)
        ph, st = [q for q in phrases if known(q)], [q for q in sets if known(q)]
        nt_ = [c for c in nots if known(c[0] + c[1])]
        assert np.array_equal(inc.posting_count(np.arange(nd + len(sparse_terms))), ref.posting_count(np.arange(nd + len(sparse_terms))))
        for k in (10, 100):
            for rt in (S.ResultType.TopkCount, S.ResultType.Count):
                _same(inc.search_lexical_batch(inc.make_queries(ph, S.QueryType.Phrase), k, rt),
                      ref.search_lexical_batch(ref.make_queries(ph, S.QueryType.Phrase), k, rt), ("phrases", l, k, rt))
                for qt in (S.QueryType.Union, S.QueryType.Intersection):
                    _same(inc.search_lexical_batch(inc.make_queries(st, qt), k, rt), ref.search_lexical_batch(ref.make_queries(st, qt), k, rt),
                          ("sets", l, k, rt, qt))
                    _same(inc.search_lexical_batch(inc.make_queries([c[0] for c in nt_], qt, [c[1] for c in nt_]), k, rt),
                          ref.search_lexical_batch(ref.make_queries([c[0] for c in nt_], qt, [c[1] for c in nt_]), k, rt), ("NOT terms", l, k, rt, qt))
    assert int(inc.search_lexical_batch(inc.make_queries([[0, 3]], S.QueryType.Phrase), 10)[3][0]) >= 300
    # refused, and the tier stays what it was: a level other than the one just committed; docs outside the level; whole lists into a
    # tier of levels; no positions for a tier that carries them; a grown DENSE vocabulary under a tier
    o, dd, tt, pp = csr(range(nd, 8), (n_levels - 1) << 16, n_docs)
    with pytest.raises(N.SeekStormHipError):
        inc.append_sparse_level(n_levels - 2, o, dd, tt, positions=pp)
    o1, d1_, t1, p1 = csr(range(nd, 8), (n_levels - 2) << 16, n_docs)
    with pytest.raises(N.SeekStormHipError):
        inc.append_sparse_level(n_levels - 1, o1, d1_, t1, positions=p1)
    with pytest.raises(N.SeekStormHipError):
        inc.append_sparse(o, dd, tt, positions=pp)
    with pytest.raises(N.SeekStormHipError):
        inc.append_sparse_level(n_levels - 1, o, dd, tt)
    o4, d4, t4, p4 = csr(range(nd + 1), (n_levels - 1) << 16, n_docs)
    with pytest.raises(N.SeekStormHipError):
        inc.append_level(n_levels - 1, dl[(n_levels - 1) << 16:], o4, d4, t4, positions=p4)
    _same(inc.search_lexical_batch(inc.make_queries(sets, S.QueryType.Union), 10), ref.search_lexical_batch(ref.make_queries(sets, S.QueryType.Union), 10),
          "after the refusals")
    ref.close()
    inc.close()


def test_multi_field_levels_equal_the_one_shot_upload(S, O):
    """ss_bm25_append_level_fields (round 6, VERDICT r5 missing 6): an image with three indexed fields committed level by level -- two full
    levels, a partial third, the re-commit of that level with more docs and a term nobody had seen -- answers every query exactly as ONE
    ss_bm25_upload_fields of the same levels does (unions, intersections, field filters, NOT terms, TopkCount), and like the oracle"""
    from test_gpu_parity import _fields_corpus
    n_fields, boost = 3, [2.0, 1.0, 0.5]
    n_docs = 2 * 65536 + 30_000
    dfs = [52_000, 23_000, 9_000, 4_000, 1_500, 300, 40]
    dl, offs, docs, fields, tfs = _fields_corpus(O, n_docs, n_fields, dfs, 77)

    def prefix(n, n_terms):  # the shard as it stands after committing docs [0, n): entries of the first n_terms terms
        o, d, f, t = [0], [], [], []
        for term in range(n_terms):
            a, b = int(offs[term]), int(offs[term + 1])
            m = docs[a:b] < n
            d.append(docs[a:b][m]); f.append(fields[a:b][m]); t.append(tfs[a:b][m]); o.append(o[-1] + int(m.sum()))
        return np.ascontiguousarray(dl[:, :n]), np.array(o, np.uint64), np.concatenate(d), np.concatenate(f), np.concatenate(t)

    def level(lo, hi, n_terms):
        o, d, f, t = [0], [], [], []
        for term in range(n_terms):
            a, b = int(offs[term]), int(offs[term + 1])
            m = (docs[a:b] >= lo) & (docs[a:b] < hi)
            d.append(docs[a:b][m]); f.append(fields[a:b][m]); t.append(tfs[a:b][m]); o.append(o[-1] + int(m.sum()))
        return np.ascontiguousarray(dl[:, lo:hi]), np.array(o, np.uint64), np.concatenate(d), np.concatenate(f), np.concatenate(t)

    lists = [[0, 1], [1, 2, 3], [4], [0, 2, 3, 4], [2, 3], [5, 0], [1, 5, 4]]
    nots = [[], [4], [], [], [0], [], [2]]
    inc, one = S.Shard(0), S.Shard(0)
    try:
        steps = [(0, 0, 65536, 6), (1, 65536, 131072, 6), (2, 131072, 131072 + 12_000, 6), (2, 131072, n_docs, 7)]  # (level, lo, hi, terms known)
        for level_ix, lo, hi, nt in steps:
            ldl, lo_, ld, lf, lt = level(lo, hi, nt)
            inc.append_level_fields(level_ix, ldl, boost, lo_, ld, lf, lt)
            pdl, po, pd, pf, pt = prefix(hi, nt)
            one.upload_lexical_fields(hi, pdl, boost, po, pd, pf, pt)
            assert inc.fields_info()[:2] == one.fields_info()[:2]
            for qt, oop in ((S.QueryType.Union, O.OP_OR), (S.QueryType.Intersection, O.OP_AND)):
                for ff in (None, [1], [0, 2]):
                    if ff is not None and qt == S.QueryType.Union:
                        continue  # (filtered unions: their own rule, tests/test_gpu_parity.py)
                    use = [(l, n) for l, n in zip(lists, nots) if max(l + n) < nt]
                    qi = inc.make_queries([u[0] for u in use], qt, [u[1] for u in use], field_filter=ff)
                    qo = one.make_queries([u[0] for u in use], qt, [u[1] for u in use], field_filter=ff)
                    assert np.array_equal(qi.view(np.uint8), qo.view(np.uint8)), "idf differs: the two images disagree on N or df"
                    a = inc.search_lexical_batch(qi, 10, S.ResultType.TopkCount, reference_shortcuts=False)
                    b = one.search_lexical_batch(qo, 10, S.ResultType.TopkCount, reference_shortcuts=False)
                    for x, y in zip(a, b):
                        assert np.array_equal(x, y), (level_ix, hi, qt, ff)
                    if ff is None:
                        for j, (terms, neg) in enumerate(use[:3]):
                            od, os_, otot, _ = O.search_fields_exhaustive(hi, pdl, boost, po, pd, pf, pt, terms, oop if len(terms) > 1 else O.OP_OR, 10, neg)
                            assert int(a[3][j]) == otot and int(a[2][j]) == len(od) and np.allclose(a[1][j][:len(od)], os_, rtol=1e-4), (level_ix, hi, qt, terms)
        # level rules: a gap, a partial level that is not the last, a change of the field count
        from seekstorm_amd import _native as N
        ldl, lo_, ld, lf, lt = level(0, 1000, 6)
        with pytest.raises(N.SeekStormHipError):
            inc.append_level_fields(5, ldl, boost, lo_, ld, lf, lt)
        with pytest.raises(N.SeekStormHipError):
            one.append_level_fields(0, ldl, boost, lo_, ld, lf, lt)  # an image that was uploaded whole: SS_ESTATE
    finally:
        inc.close(); one.close()
