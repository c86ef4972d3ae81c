"""GPU parity (-m gpu) of phrase queries (QueryType::Phrase; phrase check add_result.rs:3586-3684, positions add_result.rs:38-59,
2036-2197): docs holding every unique term whose positions carry the words consecutively, scored like the intersection,
counted only on a match -- against the oracle (reference loop restated + the definition), 2- to 6-word phrases, repeated
words, Count / Topk / TopkCount, tombstones, mixed batches."""
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


def _corpus(O, n_docs, dfs, seed, plant):
    """postings with positions; `plant` = list of (terms in order, n docs): phrases written into docs on purpose"""
    rng = np.random.default_rng(seed)
    dl = O.lex_doclen(n_docs)
    L = O.lib()
    dec = np.array([L.so_byte4_to_int(i) for i in range(256)], np.int64)
    pos_of = [dict() for _ in dfs]  # term -> doc -> set of positions
    for t, df in enumerate(dfs):
        for d in rng.choice(n_docs, df, replace=False):
            n = int(max(dec[dl[d]], 4))
            tf = int(min(rng.geometric(0.45), n // 2 + 1, 30))
            pos_of[t][int(d)] = set(int(x) for x in rng.choice(n, tf, replace=False))
    for terms, nd in plant:
        for d in rng.choice(n_docs, nd, replace=False):
            n = int(max(dec[dl[d]], len(terms) + 2))
            st = int(rng.integers(0, n - len(terms)))
            for i, t in enumerate(terms):
                pos_of[t].setdefault(int(d), set()).add(st + i)
    offs, docs, tfs, positions = [0], [], [], []
    for t in range(len(dfs)):
        ds = sorted(pos_of[t])
        for d in ds:
            ps = sorted(pos_of[t][d])
            docs.append(d); tfs.append(len(ps)); positions += ps
        offs.append(len(docs))
    return (dl, np.asarray(offs, np.uint64), np.asarray(docs, np.uint32), np.asarray(tfs, np.uint16), np.asarray(positions, np.uint16))


def test_phrase_queries_match_oracle(S, O):
    n_docs = 150_000
    dfs = [30_000, 22_000, 40_000, 9_000, 15_000, 500]
    plant = [([0, 1], 400), ([0, 1, 2], 150), ([2, 0, 2], 120), ([3, 4], 60), ([0, 1, 2, 3, 4, 5], 30), ([1, 1], 80), ([4, 0, 1], 70)]
    dl, offs, docs, tfs, positions = _corpus(O, n_docs, dfs, 5, plant)
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, offs, docs, tfs, positions)
    osh = O.Shard(n_docs, dl, offs, docs, tfs)
    osh.set_positions(positions)
    phrases = [[0, 1], [1, 0], [0, 1, 2], [2, 0, 2], [3, 4], [0, 1, 2, 3, 4, 5], [1, 1], [4, 0, 1], [5, 3], [2, 2, 2], [0, 2, 1, 0]]
    q = sh.make_queries(phrases, S.QueryType.Phrase)
    gone = list(range(3, n_docs, 97))
    for deleted in (False, True):
        sh.set_deleted(gone if deleted else [])
        osh.set_deleted(gone if deleted else [])
        for k in (10, 100):
            res = {rt: sh.search_lexical_batch(q, k, rt) for rt in (S.ResultType.TopkCount, S.ResultType.Topk, S.ResultType.Count)}
            for i, ph in enumerate(phrases):
                uniq = list(dict.fromkeys(ph))
                seq = [uniq.index(w) for w in ph]
                od, os_, otot = osh.search_phrase(uniq, seq, k, reference_loop=True)
                od2, os2, otot2 = osh.search_phrase(uniq, seq, k, reference_loop=False)
                assert otot == otot2 and np.array_equal(od, od2)  # the restated loop and the definition agree on the corpus
                doc, score, cnt, tot = res[S.ResultType.TopkCount]
                assert int(tot[i]) == otot, (ph, deleted, int(tot[i]), otot)
                assert cnt[i] == len(od) and np.allclose(score[i][:cnt[i]], os_, rtol=1e-4)
                if len(od):
                    band = abs(float(os_[-1])) * 2e-4
                    clear = lambda dd, ss: {int(x) for x, y in zip(dd, ss) if y > os_[-1] + band}
                    assert clear(doc[i][:cnt[i]], score[i][:cnt[i]]) <= set(od.tolist()) and clear(od, os_) <= set(doc[i][:cnt[i]].tolist())
                d2, s2, c2, _ = res[S.ResultType.Topk]
                assert c2[i] == cnt[i] and np.array_equal(s2[i], score[i]) and np.array_equal(d2[i], doc[i])
                assert int(res[S.ResultType.Count][3][i]) == otot and res[S.ResultType.Count][2][i] == 0
    sh.set_deleted([])
    osh.set_deleted([])
    # a phrase is stricter than the intersection of its words, and a planted phrase is found
    qi = sh.make_queries([[0, 1]], S.QueryType.Intersection)
    ti = sh.search_lexical_batch(qi, 10)[3][0]
    tp = sh.search_lexical_batch(sh.make_queries([[0, 1]], S.QueryType.Phrase), 10)[3][0]
    assert 400 <= tp < ti
    # mixed batch through the mirror: phrase and set queries side by side
    qm = sh.make_queries([[0, 1], [0, 1], [2, 0, 2]], [S.QueryType.Phrase, S.QueryType.Union, S.QueryType.Phrase])
    dm, sm, cm, tm = sh.search_lexical_batch(qm, 10)
    assert int(tm[0]) == tp and int(tm[1]) == osh.search_exhaustive([0, 1], O.OP_OR, 10)[2]
    assert int(tm[2]) == osh.search_phrase([2, 0], [0, 1, 0], 10)[2]
    # the raw ABI refuses what it does not offer
    from seekstorm_amd import _native as N
    # NOT terms with a phrase (not_query_list applies to every query type, add_result.rs:3440-3497): the phrase's matches minus the
    # docs of the NOT lists -- against the oracle's full match list
    for ph, neg in (([0, 1], [2]), ([0, 1, 2], [3]), ([2, 0, 2], [1, 4]), ([3, 4], [0])):
        uniq = list(dict.fromkeys(ph))
        od, os_, otot = osh.search_phrase(uniq, [uniq.index(w) for w in ph], n_docs)
        gone_docs = set()
        for t in neg:
            gone_docs |= set(docs[int(offs[t]):int(offs[t + 1])].tolist())
        keep = [i for i, d in enumerate(od.tolist()) if d not in gone_docs]
        for rt in (S.ResultType.TopkCount, S.ResultType.Count):
            doc, score, cnt, tot = sh.search_lexical_batch(sh.make_queries([ph], S.QueryType.Phrase, [neg]), 10, rt)
            assert int(tot[0]) == len(keep) < otot, (ph, neg, int(tot[0]), len(keep), otot)
            if rt == S.ResultType.TopkCount:
                assert cnt[0] == min(10, len(keep)) and np.allclose(score[0][:cnt[0]], os_[keep[:10]], rtol=1e-4)
                assert not set(doc[0][:cnt[0]].tolist()) & gone_docs
    sh2 = S.Shard(0)
    sh2.upload_lexical(n_docs, dl, offs, docs, tfs)  # no positions
    with pytest.raises(S.SeekStormHipError):
        sh2.search_lexical_batch(sh2.make_queries([[0, 1]], S.QueryType.Phrase), 10)
    sh2.close()
    sh.close()


def test_phrase_queries_on_an_image_built_from_index_bin(S, O):
    """the positions decoded from the file's rank/position pointers and VINT records (ss_bm25_upload_index_bin_positions):
    phrase queries on that image answer exactly like the image uploaded from arrays with the same positions, and like the
    oracle; a key head / pointer mix with embedded (<= 4 positions) and recorded postings"""
    from oracle import ref_format as RF
    n_docs = 140_000
    dfs = [30_000, 12_000, 20_000, 3_000]
    plant = [([0, 1], 300), ([0, 1, 2], 100), ([3, 0], 50), ([2, 2], 60)]
    dl, offs, docs, tfs, positions = _corpus(O, n_docs, dfs, 11, plant)
    terms, at = [], 0
    for t in range(len(dfs)):
        a, b = int(offs[t]), int(offs[t + 1])
        pl = []
        for i in range(a, b):
            pl.append(positions[at:at + int(tfs[i])].tolist())
            at += int(tfs[i])
        terms.append((1000 * (t + 1) * 8, docs[a:b].astype(np.int64), tfs[a:b].astype(np.int64), pl))  # key_hash & 7 == 0: SingleTerm
    rng = np.random.default_rng(3)
    data = RF.write_index_bin(n_docs, dl, terms, rng)
    ix = S.IndexBin(data)
    a, b = S.Shard(0), S.Shard(0)
    a.upload_index_bin(ix, positions=True)
    b.upload_lexical(n_docs, dl, offs, docs, tfs, positions)
    osh = O.Shard(n_docs, dl, offs, docs, tfs)
    osh.set_positions(positions)
    phrases = [[0, 1], [0, 1, 2], [3, 0], [2, 2], [1, 0], [2, 0, 1]]
    for k in (10, 100):
        ra = a.search_lexical_batch(a.make_queries(phrases, S.QueryType.Phrase), k)
        rb = b.search_lexical_batch(b.make_queries(phrases, S.QueryType.Phrase), k)
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y)
        for i, ph in enumerate(phrases):
            uniq = list(dict.fromkeys(ph))
            od, os_, otot = osh.search_phrase(uniq, [uniq.index(w) for w in ph], k)
            assert int(ra[3][i]) == otot and ra[2][i] == len(od) and np.allclose(ra[1][i][:len(od)], os_, rtol=1e-4)
    # without the positions the same file still serves set queries, and refuses phrases
    a.upload_index_bin(ix)
    with pytest.raises(S.SeekStormHipError):
        a.search_lexical_batch(a.make_queries([[0, 1]], S.QueryType.Phrase), 10)
    a.close()
    b.close()


def test_phrase_queries_on_a_default_index_with_ngram_keys(S, O):
    """The reference's DEFAULT index (NgramFF | NgramFFF, index.rs:1422-1424; 23-byte key heads): sequences of frequent terms are
    indexed as n-gram keys beside their single terms, and a phrase query that names such a sequence resolves to the KEY -- one
    entry whose positions are those of its first word and whose successor stands 2 / 3 places later (search.rs:3305-3328).
    ss_bm25_upload_index_bin_positions puts the key's positions behind its first component term; phrases mixing keys and single
    terms answer like the oracle (so_search_phrase_items: same docs as the phrase over the single terms, scored by the n-gram arm)."""
    import ngram_corpus as NG
    from oracle import ref_format as RF
    from seekstorm_amd.search import idf_f32
    n_docs = 140_000
    C = NG.build(O, _corpus, n_docs, [30_000, 12_000, 20_000, 3_000],
                 [([0, 1], 300), ([0, 1, 2], 200), ([3, 0, 1], 80), ([0, 1, 3], 90), ([0, 1, 2, 3], 40), ([2, 0, 1], 70), ([0, 1, 0, 1], 25), ([3, 0, 1, 2], 35)], 12)
    data = RF.write_index_bin(n_docs, C.dl, C.terms, np.random.default_rng(3), key_head_size=23, ngram_terms=C.ngram_terms)
    ix = S.IndexBin(data, key_head_size=23)
    a = S.Shard(0)
    a.upload_index_bin(ix, positions=True)
    tid = {t: ix.term_of_key(NG.KEY(t)) for t in range(4)}
    ent, idf_of = {}, {}
    for words, key in NG.KEYS.items():
        comp = ix.terms_of_key(key)
        assert len(comp) == len(words)
        ent[words] = tuple(t for t, _ in comp)
        idf_of.update({t: i for t, i in comp})
    osh = C.oracle_shard(O)
    gq = a.make_queries([[ent[e] if isinstance(e, tuple) else tid[e] for e in ph] for ph in NG.PHRASES], S.QueryType.Phrase, idf_of=idf_of)
    sq = a.make_queries([[tid[w] for w in ph] for ph in NG.SAME_DOCS_AS], S.QueryType.Phrase)
    for k in (10, 100):
        for rt in (S.ResultType.TopkCount, S.ResultType.Count):
            rg = a.search_lexical_batch(gq, k, rt)
            rs = a.search_lexical_batch(sq, k, rt)
            for i, ph in enumerate(NG.PHRASES):
                uniq, seq, places, idf = C.oracle_query(ph, lambda e, c: idf_of[ent[e][c]], lambda l: float(idf_f32(n_docs, osh.df(l))))
                od, os_, otot = osh.search_phrase_items(uniq, seq, places, k, idf=idf, reference_loop=True)
                assert otot > 0, ph
                assert int(rg[3][i]) == otot, (ph, int(rg[3][i]), otot)
                assert int(rs[3][i]) == otot, ("the phrase over the single terms matches the same docs", ph)
                if rt == S.ResultType.TopkCount:
                    assert rg[2][i] == len(od) and np.allclose(rg[1][i][:len(od)], os_, rtol=1e-4), ph
                    band = abs(float(os_[-1])) * 2e-4
                    clear = lambda dd, ss: {int(x) for x, y in zip(dd, ss) if y > os_[-1] + band}
                    assert clear(rg[0][i][:rg[2][i]], rg[1][i][:rg[2][i]]) <= set(od.tolist()) and clear(od, os_) <= set(rg[0][i][:rg[2][i]].tolist())
    # a phrase that is nothing but ONE key is a term query over the key (search.rs:3544): every doc of the key, no position check
    one = a.search_lexical_batch(a.make_queries([[ent[NG.AB]]], S.QueryType.Phrase, idf_of=idf_of), 10)
    assert int(one[3][0]) == len(C.rows_of[NG.AB])
    a.close()


def test_positions_beside_a_sparse_tier(S, O):
    """ss_index_bin_tier + ss_bm25_upload_index_bin_positions: the dense terms carry positions (phrases over them answer like the
    untiered image), the rare keys sit in the sparse tier with THEIR positions: set queries and phrases naming the rare word"""
    import ngram_corpus as NG
    from oracle import ref_format as RF
    n_docs = 70_000
    C = NG.build(O, _corpus, n_docs, [9_000, 5_000, 7_000, 60], [([0, 1], 200), ([0, 1, 2], 100), ([3, 0], 20)], 31)
    data = RF.write_index_bin(n_docs, C.dl, C.terms, np.random.default_rng(4), key_head_size=23, ngram_terms=C.ngram_terms)
    ix_t, ix_u = S.IndexBin(data, key_head_size=23), S.IndexBin(data, key_head_size=23)
    n_dense = ix_t.tier(1000)  # term 3 (df ~ 80) and the keys' lists (a few hundred docs) go to the sparse tier
    assert 0 < n_dense < ix_t.term_count
    a, b = S.Shard(0), S.Shard(0)
    a.upload_index_bin(ix_t, positions=True)
    b.upload_index_bin(ix_u, positions=True)
    ta = {t: ix_t.term_of_key(NG.KEY(t)) for t in range(4)}
    tb = {t: ix_u.term_of_key(NG.KEY(t)) for t in range(4)}
    assert ta[3] >= n_dense and max(ta[0], ta[1], ta[2]) < n_dense
    for ph in ([0, 1], [0, 1, 2], [1, 0], [2, 0, 1]):
        ra = a.search_lexical_batch(a.make_queries([[ta[w] for w in ph]], S.QueryType.Phrase), 10)
        rb = b.search_lexical_batch(b.make_queries([[tb[w] for w in ph]], S.QueryType.Phrase), 10)
        assert int(ra[3][0]) == int(rb[3][0]) and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[0], rb[0]), ph
    ua = a.search_lexical_batch(a.make_queries([[ta[3], ta[0]]], S.QueryType.Union), 10)
    ub = b.search_lexical_batch(b.make_queries([[tb[3], tb[0]]], S.QueryType.Union), 10)
    assert int(ua[3][0]) == int(ub[3][0]) and np.allclose(ua[1], ub[1], rtol=1e-6)
    # a phrase naming the rare word: driven by its sparse list, positions of both tiers -- the untiered image's answer
    for ph in ([3, 0], [0, 3], [3, 0, 1], [3, 3]):
        ra = a.search_lexical_batch(a.make_queries([[ta[w] for w in ph]], S.QueryType.Phrase), 10)
        rb = b.search_lexical_batch(b.make_queries([[tb[w] for w in ph]], S.QueryType.Phrase), 10)
        assert int(ra[3][0]) == int(rb[3][0]) and np.array_equal(ra[2], rb[2]) and np.allclose(ra[1], rb[1], rtol=1e-6), ph
        n = int(ra[2][0])
        assert set(ra[0][0][:n].tolist()) == set(rb[0][0][:n].tolist())
    assert int(a.search_lexical_batch(a.make_queries([[ta[3], ta[0]]], S.QueryType.Phrase), 10)[3][0]) >= 20  # the planted ones
    a.close()
    b.close()


def test_phrases_naming_sparse_terms(S, O):
    """phrases whose rare words sit in the SPARSE tier (ss_bm25_append_sparse_positions): the shortest sparse list drives, every other
    word is found by binary search with its positions (sparse: the tier's pool; dense: the image's) -- 2- to 5-word phrases, repeated
    words, NOT terms of either tier, tombstones, counts, k = 10 / 100, mixed batches; against the oracle holding every list"""
    from seekstorm_amd import _native as N
    n_docs = 120_000
    dfs = [30_000, 22_000, 9_000, 500, 300, 1_500, 40]
    nd = 3
    plant = [([0, 3], 80), ([3, 4], 40), ([1, 3, 2], 60), ([5, 0, 5], 50), ([4, 4], 30), ([0, 1], 300), ([6, 5, 3, 0, 1], 12), ([2, 5], 70)]
    dl, offs, docs, tfs, positions = _corpus(O, n_docs, dfs, 23, plant)
    e = int(offs[nd])
    pe = int(tfs[:e].astype(np.int64).sum())
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, offs[:nd + 1], docs[:e], tfs[:e], positions[:pe])
    mid = nd + 2
    m = int(offs[mid]); pm = int(tfs[:m].astype(np.int64).sum())
    assert sh.append_sparse(offs[nd:mid + 1] - offs[nd], docs[e:m], tfs[e:m], positions=positions[pe:pm]) == nd
    assert sh.append_sparse(offs[mid:] - offs[mid], docs[m:], tfs[m:], positions=positions[pm:]) == mid
    osh = O.Shard(n_docs, dl, offs, docs, tfs)
    osh.set_positions(positions)
    cases = [([0, 3], []), ([3, 0], []), ([3, 4], []), ([1, 3, 2], []), ([5, 0, 5], []), ([4, 4], []), ([6, 5, 3, 0, 1], []), ([2, 5], []),
             ([0, 3], [1]), ([2, 5], [3, 0]), ([3, 4], [5]), ([0, 1], [3]), ([0, 1], []), ([5, 2], [])]
    q = sh.make_queries([c[0] for c in cases], S.QueryType.Phrase, [c[1] for c in cases])
    gone = list(range(3, n_docs, 61))
    for deleted in (False, True):
        sh.set_deleted(gone if deleted else [])
        osh.set_deleted(gone if deleted else [])
        for k in (10, 100):
            res = {rt: sh.search_lexical_batch(q, k, rt) for rt in (S.ResultType.TopkCount, S.ResultType.Count)}
            for i, (ph, neg) in enumerate(cases):
                uniq = list(dict.fromkeys(ph))
                seq = [uniq.index(w) for w in ph]
                od, os_, otot = osh.search_phrase(uniq, seq, n_docs)
                drop = set()
                for t in neg:
                    drop |= set(docs[int(offs[t]):int(offs[t + 1])].tolist())
                keep = [j for j, d in enumerate(od.tolist()) if d not in drop]
                od, os_ = od[keep][:k], os_[keep][:k]
                doc, score, cnt, tot = res[S.ResultType.TopkCount]
                assert int(tot[i]) == len(keep), (ph, neg, deleted, k, int(tot[i]), len(keep))
                assert cnt[i] == len(od) and np.allclose(score[i][:cnt[i]], os_, rtol=1e-4), (ph, neg)
                if len(od) < k:
                    assert set(doc[i][:cnt[i]].tolist()) == set(od.tolist())
                assert int(res[S.ResultType.Count][3][i]) == len(keep)
    sh.set_deleted([])
    osh.set_deleted([])
    assert int(sh.search_lexical_batch(sh.make_queries([[0, 3]], S.QueryType.Phrase), 10)[3][0]) >= 80
    # a batch mixing sparse phrases with set queries and dense phrases
    qm = sh.make_queries([[0, 3], [0, 3], [0, 1], [3, 4], [3, 0]], [S.QueryType.Phrase, S.QueryType.Union, S.QueryType.Phrase, S.QueryType.Intersection,
                                                                    S.QueryType.Phrase])
    tm = sh.search_lexical_batch(qm, 10, reference_shortcuts=False)[3]
    assert int(tm[0]) == osh.search_phrase([0, 3], [0, 1], 10)[2] and int(tm[1]) == osh.search_exhaustive([0, 3], O.OP_OR, 10)[2]
    assert int(tm[2]) == osh.search_phrase([0, 1], [0, 1], 10)[2] and int(tm[3]) == osh.search_exhaustive([3, 4], O.OP_AND, 10)[2]
    assert int(tm[4]) == osh.search_phrase([3, 0], [0, 1], 10)[2]
    # a tier without positions refuses the phrase, not the set query
    sh2 = S.Shard(0)
    sh2.upload_lexical(n_docs, dl, offs[:nd + 1], docs[:e], tfs[:e], positions[:pe])
    sh2.append_sparse(offs[nd:] - offs[nd], docs[e:], tfs[e:])
    with pytest.raises(N.SeekStormHipError):
        sh2.search_lexical_batch(sh2.make_queries([[0, 3]], S.QueryType.Phrase), 10)
    sh2.search_lexical_batch(sh2.make_queries([[0, 3]], S.QueryType.Intersection), 10)
    sh2.close()
    sh.close()


# ------------------------------------------------------------------------------------------------ several indexed fields
def _corpus_fields(O, n_docs, n_fields, dfs, seed, plant, cross):
    """(term, doc, field) entries with positions.  plant = [(words, field, n docs)]: the phrase written into that field;
    cross = [(word a, word b, n docs)]: a at the LAST position of field 0 and b at position 0 of field 1, and a at p in field 0 with
    b at p + 1 in field 2 -- consecutive numbers in different fields, which are no phrase"""
    rng = np.random.default_rng(seed)
    dl = np.stack([O.lex_doclen(n_docs, seed=O.LEX_SEED + 7 * f) for f in range(n_fields)])
    pos_of = [dict() for _ in dfs]  # term -> (doc, field) -> set of positions
    for t, df in enumerate(dfs):
        for d in rng.choice(n_docs, df, replace=False):
            k = int(rng.integers(1, n_fields + 1)) if rng.random() < 0.4 else 1
            for f in rng.choice(n_fields, size=k, replace=False):
                tf = int(min(rng.geometric(0.5), 12))
                pos_of[t][(int(d), int(f))] = set(int(x) for x in rng.choice(60, tf, replace=False))
    for words, f, nd in plant:
        for d in rng.choice(n_docs, nd, replace=False):
            st = int(rng.integers(0, 50))
            for i, t in enumerate(words):
                pos_of[t].setdefault((int(d), f), set()).add(st + i)
    for a, b, nd in cross:
        for d in rng.choice(n_docs, nd, replace=False):
            pos_of[a].setdefault((int(d), 0), set()).add(65535)
            pos_of[b].setdefault((int(d), 1), set()).add(0)
            pos_of[a].setdefault((int(d), 0), set()).add(200)
            pos_of[b].setdefault((int(d), 2), set()).add(201)
    offs, docs, fields, tfs, positions = [0], [], [], [], []
    for t in range(len(dfs)):
        for (d, f) in sorted(pos_of[t]):
            ps = sorted(pos_of[t][(d, f)])
            docs.append(d); fields.append(f); tfs.append(len(ps)); positions += ps
        offs.append(len(docs))
    return (dl, np.asarray(offs, np.uint64), np.asarray(docs, np.uint32), np.asarray(fields, np.uint8), np.asarray(tfs, np.uint16),
            np.asarray(positions, np.uint16))


def test_phrase_queries_over_several_indexed_fields(S, O):
    """add_result_multiterm_multifield's phrase check (add_result.rs:3248-3386): the phrase must stand inside ONE field, fields in
    ascending order, only listed fields under a field filter; score = BM25F over all fields of the unique terms.  The HIP path reads
    the merged lists and their field-tagged positions (ss_bm25_upload_fields_positions)."""
    n_docs, n_fields = 120_000, 3
    dfs = [26_000, 20_000, 33_000, 8_000, 12_000, 400]
    plant = [([0, 1], 0, 300), ([0, 1], 2, 200), ([0, 1, 2], 1, 150), ([2, 0, 2], 0, 100), ([3, 4], 2, 60), ([0, 1, 2, 3, 4, 5], 1, 30),
             ([1, 1], 2, 80), ([4, 0, 1], 0, 70)]
    cross = [(0, 1, 500), (3, 4, 300)]
    boost = np.array([2.0, 1.0, 0.5], np.float32)
    dl, offs, docs, fields, tfs, positions = _corpus_fields(O, n_docs, n_fields, dfs, 17, plant, cross)
    sh = S.Shard(0)
    sh.upload_lexical_fields(n_docs, dl, boost, offs, docs, fields, tfs, positions)
    phrases = [[0, 1], [1, 0], [0, 1, 2], [2, 0, 2], [3, 4], [0, 1, 2, 3, 4, 5], [1, 1], [4, 0, 1], [5, 3], [2, 2, 2]]
    gone = list(range(5, n_docs, 89))
    for filt in ((), (0,), (1, 2), (2,)):
        q = sh.make_queries(phrases, S.QueryType.Phrase, field_filter=filt)
        for deleted in (False, True):
            sh.set_deleted(gone if deleted else [])
            for k in (10, 100):
                res = {rt: sh.search_lexical_batch(q, k, rt) for rt in (S.ResultType.TopkCount, S.ResultType.Topk, S.ResultType.Count)}
                for i, ph in enumerate(phrases):
                    uniq = list(dict.fromkeys(ph))
                    seq = [uniq.index(w) for w in ph]
                    args = (n_docs, dl, boost, offs, docs, fields, tfs, positions, uniq, seq, k)
                    od, os_, otot = O.search_fields_phrase(*args, deleted=gone if deleted else (), field_filter=filt, reference_loop=True)
                    od2, os2, otot2 = O.search_fields_phrase(*args, deleted=gone if deleted else (), field_filter=filt, reference_loop=False)
                    assert otot == otot2 and np.array_equal(od, od2)
                    doc, score, cnt, tot = res[S.ResultType.TopkCount]
                    assert int(tot[i]) == otot, (ph, filt, deleted, int(tot[i]), otot)
                    assert cnt[i] == len(od) and np.allclose(score[i][:cnt[i]], os_, rtol=1e-4), (ph, filt)
                    if len(od):
                        band = abs(float(os_[-1])) * 2e-4
                        clear = lambda dd, ss: {int(x) for x, y in zip(dd, ss) if y > os_[-1] + band}
                        assert clear(doc[i][:cnt[i]], score[i][:cnt[i]]) <= set(od.tolist()) and clear(od, os_) <= set(doc[i][:cnt[i]].tolist())
                    d2, s2, c2, _ = res[S.ResultType.Topk]
                    assert c2[i] == cnt[i] and np.array_equal(s2[i], score[i]) and np.array_equal(d2[i], doc[i])
                    assert int(res[S.ResultType.Count][3][i]) == otot and res[S.ResultType.Count][2][i] == 0
    sh.set_deleted([])
    # the planted cross-field neighbours are no phrase: [3, 4] matches only where it was planted inside a field (or by chance)
    t_all = int(sh.search_lexical_batch(sh.make_queries([[3, 4]], S.QueryType.Phrase), 10)[3][0])
    t_and = int(sh.search_lexical_batch(sh.make_queries([[3, 4]], S.QueryType.Intersection), 10)[3][0])
    assert 60 <= t_all < t_and - 250
    # a filter can only remove matches; the fields partition them at most (a doc may carry the phrase in two fields)
    per = [int(sh.search_lexical_batch(sh.make_queries([[0, 1]], S.QueryType.Phrase, field_filter=(f,)), 10)[3][0]) for f in range(3)]
    whole = int(sh.search_lexical_batch(sh.make_queries([[0, 1]], S.QueryType.Phrase), 10)[3][0])
    assert max(per) <= whole <= sum(per) and per[0] >= 300 and per[2] >= 200
    # mixed batch: a phrase, a union and a filtered intersection side by side
    qm = sh.make_queries([[0, 1], [0, 1], [2, 0, 2]], [S.QueryType.Phrase, S.QueryType.Union, S.QueryType.Phrase])
    tm = sh.search_lexical_batch(qm, 10)[3]
    assert int(tm[0]) == whole and int(tm[1]) == O.search_fields_exhaustive(n_docs, dl, boost, offs, docs, fields, tfs, [0, 1], O.OP_OR, 10)[2]
    # an image without positions refuses phrases
    sh.upload_lexical_fields(n_docs, dl, boost, offs, docs, fields, tfs)
    with pytest.raises(S.SeekStormHipError):
        sh.search_lexical_batch(sh.make_queries([[0, 1]], S.QueryType.Phrase), 10)
    # a positions array of the wrong length is refused before anything is built
    with pytest.raises(S.SeekStormHipError):
        sh.upload_lexical_fields(n_docs, dl, boost, offs, docs, fields, tfs, positions[:-1])
    sh.close()


def test_phrases_naming_sparse_terms_over_several_indexed_fields(S, O):
    """the sparse tier of a multi-field image with positions (ss_bm25_append_sparse_fields_positions): merged lists, field-tagged
    positions; the phrase must stand inside one field, a field filter lists the fields it may stand in -- against the BM25F phrase oracle"""
    n_docs, n_fields = 90_000, 3
    dfs = [26_000, 20_000, 8_000, 400, 900, 150]
    nd = 3
    plant = [([0, 3], 0, 90), ([0, 3], 2, 60), ([3, 4], 1, 50), ([1, 4, 2], 2, 40), ([5, 0, 5], 0, 30), ([3, 3], 1, 25), ([0, 1], 0, 200)]
    cross = [(0, 3, 200), (3, 4, 100)]
    boost = np.array([2.0, 1.0, 0.5], np.float32)
    dl, offs, docs, fields, tfs, positions = _corpus_fields(O, n_docs, n_fields, dfs, 29, plant, cross)
    e = int(offs[nd]); pe = int(tfs[:e].astype(np.int64).sum())
    sh = S.Shard(0)
    sh.upload_lexical_fields(n_docs, dl, boost, offs[:nd + 1], docs[:e], fields[:e], tfs[:e], positions[:pe])
    assert sh.append_sparse_fields(offs[nd:] - offs[nd], docs[e:], fields[e:], tfs[e:], positions=positions[pe:]) == nd
    phrases = [[0, 3], [3, 0], [3, 4], [1, 4, 2], [5, 0, 5], [3, 3], [4, 3], [0, 1]]
    gone = list(range(5, n_docs, 89))
    for filt in ((), (0,), (1, 2)):
        q = sh.make_queries(phrases, S.QueryType.Phrase, field_filter=filt)
        for deleted in (False, True):
            sh.set_deleted(gone if deleted else [])
            for k in (10, 100):
                res = {rt: sh.search_lexical_batch(q, k, rt) for rt in (S.ResultType.TopkCount, S.ResultType.Count)}
                for i, ph in enumerate(phrases):
                    uniq = list(dict.fromkeys(ph))
                    seq = [uniq.index(w) for w in ph]
                    od, os_, otot = O.search_fields_phrase(n_docs, dl, boost, offs, docs, fields, tfs, positions, uniq, seq, k,
                                                           deleted=gone if deleted else (), field_filter=filt, reference_loop=False)
                    doc, score, cnt, tot = res[S.ResultType.TopkCount]
                    assert int(tot[i]) == otot, (ph, filt, deleted, int(tot[i]), otot)
                    assert cnt[i] == len(od) and np.allclose(score[i][:cnt[i]], os_, rtol=1e-4), (ph, filt)
                    if len(od) < k:
                        assert set(doc[i][:cnt[i]].tolist()) == set(od.tolist())
                    assert int(res[S.ResultType.Count][3][i]) == otot
    sh.set_deleted([])
    # the cross-field neighbours are no phrase
    t_ph = int(sh.search_lexical_batch(sh.make_queries([[0, 3]], S.QueryType.Phrase), 10)[3][0])
    t_and = int(sh.search_lexical_batch(sh.make_queries([[0, 3]], S.QueryType.Intersection), 10, reference_shortcuts=False)[3][0])
    assert 150 <= t_ph <= t_and - 150
    sh.close()


def test_phrase_queries_on_a_multi_field_index_bin(S, O):
    """positions of a multi-field index.bin (field vectors + VINT positions in the records, bit-packed positions in the embedded
    pointers; add_result.rs:1485-2034): the image built from the file answers phrase queries exactly like the image uploaded from
    the arrays, and like the oracle"""
    from oracle import ref_format as RF
    n_docs, n_fields, longest = 100_000, 3, 1
    dfs = [22_000, 9_000, 15_000, 2_500]
    plant = [([0, 1], 0, 250), ([0, 1, 2], 1, 100), ([3, 0], 2, 50), ([2, 2], 1, 60)]
    dl, offs, docs, fields, tfs, positions = _corpus_fields(O, n_docs, n_fields, dfs, 23, plant, [(0, 1, 200)])
    terms, at = [], 0
    for t in range(len(dfs)):
        a, b = int(offs[t]), int(offs[t + 1])
        pl = []
        for i in range(a, b):
            pl.append(positions[at:at + int(tfs[i])].tolist())
            at += int(tfs[i])
        terms.append((1000 * (t + 1) * 8, docs[a:b].astype(np.int64), fields[a:b].astype(np.int64), tfs[a:b].astype(np.int64), pl))
    rng = np.random.default_rng(4)
    data = RF.write_index_bin(n_docs, dl, terms, rng, n_fields=n_fields, longest_field_id=longest)
    ix = S.IndexBin(data, n_fields)
    boost = np.array([1.5, 1.0, 0.75], np.float32)
    a, b = S.Shard(0), S.Shard(0)
    a.upload_index_bin(ix, boost=boost, positions=True)
    b.upload_lexical_fields(n_docs, dl, boost, offs, docs, fields, tfs, positions)
    phrases = [[0, 1], [0, 1, 2], [3, 0], [2, 2], [1, 0], [2, 0, 1]]
    for filt in ((), (1,), (0, 2)):
        for k in (10, 100):
            ra = a.search_lexical_batch(a.make_queries(phrases, S.QueryType.Phrase, field_filter=filt), k)
            rb = b.search_lexical_batch(b.make_queries(phrases, S.QueryType.Phrase, field_filter=filt), k)
            for x, y in zip(ra, rb):
                assert np.array_equal(x, y)
            for i, ph in enumerate(phrases):
                uniq = list(dict.fromkeys(ph))
                od, os_, otot = O.search_fields_phrase(n_docs, dl, boost, offs, docs, fields, tfs, positions, uniq, [uniq.index(w) for w in ph], k,
                                                       field_filter=filt)
                assert int(ra[3][i]) == otot and ra[2][i] == len(od) and np.allclose(ra[1][i][:len(od)], os_, rtol=1e-4), (ph, filt)
    assert int(a.search_lexical_batch(a.make_queries([[0, 1]], S.QueryType.Phrase), 10)[3][0]) >= 250
    a.close()
    b.close()


def test_phrase_queries_over_ngram_keys_of_a_multi_field_index(S, O):
    """a DEFAULT index with several indexed fields: an n-gram key's record carries the field vector of every component term, then the
    key's own field vector and positions (index_posting.rs:666-741; reader add_result.rs:1524-1600).  The key's positions go behind its
    first component term, field by field; phrases whose entries are keys answer like the oracle (so_search_fields_phrase_items) and
    match the docs the phrase over the single terms matches, with and without a field filter."""
    from oracle import ref_format as RF
    from seekstorm_amd.search import idf_f32
    n_docs, n_fields, longest = 80_000, 3, 1
    dfs = [18_000, 8_000, 12_000, 2_500]
    plant = [([0, 1], 0, 250), ([0, 1, 2], 1, 120), ([3, 0, 1], 2, 60), ([0, 1, 3], 1, 70), ([2, 0, 1], 0, 50), ([0, 1, 2, 3], 2, 40)]
    dl, offs, docs, fields, tfs, positions = _corpus_fields(O, n_docs, n_fields, dfs, 29, plant, [(0, 1, 150)])
    ent_pos, at = [dict() for _ in dfs], 0  # term -> (doc, field) -> positions
    for t in range(len(dfs)):
        for i in range(int(offs[t]), int(offs[t + 1])):
            ent_pos[t][(int(docs[i]), int(fields[i]))] = positions[at:at + int(tfs[i])].tolist()
            at += int(tfs[i])
    doc_fields = [dict() for _ in dfs]          # term -> doc -> [(field, tf)]
    for t in range(len(dfs)):
        for (d, f), ps in sorted(ent_pos[t].items()):
            doc_fields[t].setdefault(d, []).append((f, len(ps)))
    KEY = lambda t: 1000 * (t + 1) * 8
    keys = {(0, 1): 0x7000_0000_0000 | 1, (0, 1, 2): 0x7100_0000_0000 | 4}
    lut = lambda df: int(O.lib().so_int_to_byte4(int(df)))
    ngram_terms, rows_of = [], {}
    for words, key in keys.items():
        rows = {}  # doc -> [(field, positions of the key in the field)]
        for (d, f), ps in sorted(ent_pos[words[0]].items()):
            if all((d, f) in ent_pos[w] for w in words[1:]):
                sets = [set(ent_pos[w][(d, f)]) for w in words]
                hit = [p_ for p_ in ps if all(p_ + i in sets[i] for i in range(1, len(words)))]
                if hit:
                    rows.setdefault(d, []).append((f, hit))
        rows_of[words] = rows
        kd, kf, kc, kp, vecs = [], [], [], [], {}
        for d in sorted(rows):
            for f, hit in rows[d]:
                kd.append(d); kf.append(f); kc.append(len(hit)); kp.append(hit)
            vecs[d] = [doc_fields[w][d] for w in words]
        ngram_terms.append((key, np.array(kd, np.int64), np.array(kf, np.int64), np.array(kc, np.int64), vecs, [lut(len(doc_fields[w])) for w in words], kp))
    terms = []
    for t in range(len(dfs)):
        a, b = int(offs[t]), int(offs[t + 1])
        terms.append((KEY(t), docs[a:b].astype(np.int64), fields[a:b].astype(np.int64), tfs[a:b].astype(np.int64),
                      [ent_pos[t][(int(docs[i]), int(fields[i]))] for i in range(a, b)]))
    data = RF.write_index_bin(n_docs, dl, terms, np.random.default_rng(4), n_fields=n_fields, longest_field_id=longest, key_head_size=23,
                              ngram_terms=ngram_terms)
    ix = S.IndexBin(data, n_fields, key_head_size=23)
    boost = np.array([1.5, 1.0, 0.75], np.float32)
    a = S.Shard(0)
    a.upload_index_bin(ix, boost=boost, positions=True)
    tid = {t: ix.term_of_key(KEY(t)) for t in range(len(dfs))}
    ent, idf_of = {}, {}
    for words, key in keys.items():
        comp = ix.terms_of_key(key)
        ent[words] = tuple(t for t, _ in comp)
        idf_of.update({t: i for t, i in comp})
    # the oracle's lists: single terms, then one list per component of every key; counts = positions behind every entry
    o_offs, o_docs, o_f, o_tf, o_cnt, o_pos, o_id = [0], [], [], [], [], [], {}
    for t in range(len(dfs)):
        o_id[("t", t)] = len(o_offs) - 1
        for (d, f), ps in sorted(ent_pos[t].items()):
            o_docs.append(d); o_f.append(f); o_tf.append(len(ps)); o_cnt.append(len(ps)); o_pos += ps
        o_offs.append(len(o_docs))
    for words in keys:
        for c, w in enumerate(words):
            o_id[(words, c)] = len(o_offs) - 1
            for d in sorted(rows_of[words]):
                own = dict(rows_of[words][d])
                for f, tf in doc_fields[w][d]:
                    hit = own.get(f, []) if c == 0 else []
                    o_docs.append(d); o_f.append(f); o_tf.append(tf); o_cnt.append(len(hit)); o_pos += hit
            o_offs.append(len(o_docs))
    AB, ABC = (0, 1), (0, 1, 2)
    phrases = [[AB, 3], [3, AB], [ABC, 3], [2, AB], [AB, 2]]
    same_docs_as = [[0, 1, 3], [3, 0, 1], [0, 1, 2, 3], [2, 0, 1], [0, 1, 2]]
    for filt in ((), (1,), (0, 2)):
        gq = a.make_queries([[ent[e] if isinstance(e, tuple) else tid[e] for e in ph] for ph in phrases], S.QueryType.Phrase, idf_of=idf_of, field_filter=filt)
        sq = a.make_queries([[tid[w] for w in ph] for ph in same_docs_as], S.QueryType.Phrase, field_filter=filt)
        for k in (10, 100):
            rg = a.search_lexical_batch(gq, k)
            rs = a.search_lexical_batch(sq, k)
            for i, ph in enumerate(phrases):
                uniq, seq, places, idf, at_place = [], [], [], [], 0
                for e in ph:
                    lists = [o_id[(e, c)] for c in range(len(e))] if isinstance(e, tuple) else [o_id[("t", e)]]
                    for c, l in enumerate(lists):
                        if l not in uniq:
                            uniq.append(l)
                            n_l = len({o_docs[j] for j in range(o_offs[l], o_offs[l + 1])})
                            idf.append(idf_of[ent[e][c]] if isinstance(e, tuple) else float(idf_f32(n_docs, n_l)))
                    seq.append(uniq.index(lists[0])); places.append(at_place)
                    at_place += len(lists)
                od, os_, otot = O.search_fields_phrase_items(n_docs, dl, boost, o_offs, o_docs, o_f, o_tf, o_cnt, o_pos, uniq, seq, places, k, idf=idf,
                                                             field_filter=filt)
                assert int(rg[3][i]) == otot, (ph, filt, int(rg[3][i]), otot)
                assert int(rs[3][i]) == otot, ("the phrase over the single terms matches the same docs", ph, filt)
                assert rg[2][i] == len(od) and np.allclose(rg[1][i][:len(od)], os_, rtol=1e-4), (ph, filt)
        if not filt:
            assert all(int(x) > 0 for x in rg[3])
    a.close()


def test_phrase_queries_take_the_pool_rows_first_on_a_rationed_vocabulary(S, O):
    """Rationed probe rows (more lists than rows): the phrase kernel of the probe index needs a row for every list it reads and has no scan
    fallback -- a phrase over a row-less list is answered by the generic galloping kernel instead (bm25_gallop.hip, round 6); the other
    queries of the batch get the pool's rows, and those whose lists found none are answered by the scan kernels.  Same answers as the
    unrationed shard, bit for bit."""
    n_docs = 120_000
    dfs = [int(120_000 * 0.25 / (1 + 0.3 * i)) for i in range(48)]
    plant = [([40, 41], 200), ([42, 43, 40], 90), ([44, 45], 120), ([2, 3], 300)]
    dl, offs, docs, tfs, positions = _corpus(O, n_docs, dfs, 31, plant)
    full, part = S.Shard(0), S.Shard(0)
    full.upload_lexical(n_docs, dl, offs, docs, tfs, positions)
    n_sub = (n_docs + 4095) // 4096
    part.set_probe_budget((40 + 1) * n_sub * 64 * 12)   # 40 rows for 48 lists: 30 fixed (the longest) + a pool of 10
    part.upload_lexical(n_docs, dl, offs, docs, tfs, positions)
    assert not part.terms_probed(list(range(30, 48))).any()
    # one batch: unions over 12 row-less lists in front, then the phrases over 6 more -- 18 lists for a pool of 10
    tl = [[30, 31, 32], [33, 34, 35], [36, 37, 38], [39, 46, 47], [40, 41], [42, 43, 40], [44, 45], [2, 3]]
    qt = [S.QueryType.Union] * 4 + [S.QueryType.Phrase] * 4
    for k in (10, 100):
        a = full.search_lexical_batch(full.make_queries(tl, qt), k)
        b = part.search_lexical_batch(part.make_queries(tl, qt), k)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), k
    assert int(a[3][4]) >= 200 and int(a[3][5]) >= 90 and int(a[3][6]) >= 120
    # round 6: a phrase over a row-less list takes the generic galloping kernel (no rows needed) -- the pool's rows stay with the unions,
    # and a batch whose phrases alone name more row-less lists than the pool holds is ANSWERED (round 5 refused it)
    assert part.generic_batches() >= 2
    ph = [[30, 31, 32, 33], [34, 35, 36, 37], [38, 39, 46]]
    a = full.search_lexical_batch(full.make_queries(ph, S.QueryType.Phrase), 10)
    b = part.search_lexical_batch(part.make_queries(ph, S.QueryType.Phrase), 10)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # and a phrase over lists that all have rows keeps the probe index's kernel
    b = part.search_lexical_batch(part.make_queries([[44, 45]], S.QueryType.Phrase), 10)
    a = full.search_lexical_batch(full.make_queries([[44, 45]], S.QueryType.Phrase), 10)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and int(a[3][0]) >= 120
    full.close()
    part.close()
