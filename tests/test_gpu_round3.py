"""GPU tests (-m gpu) added in round 3: the vector and hybrid shard tasks behind the RCCL exchange (ss_vec_search_sharded,
ss_hybrid_search_sharded) with a communicator of one rank, the status agreement of the collective searches, the timing hook of
the collective, i8 Euclidean searches without their norms."""
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


def test_vector_and_hybrid_sharded_single_rank(S, O, both):
    """with ONE rank the exchange is the identity: ss_vec_search_sharded = ss_vec_search with u64 ids, ss_hybrid_search_sharded =
    Index.search(SearchMode.Hybrid) of the Python mirror (two searches + ss_merge_results on the host), totals = max(lexical,
    vector); the collective's time is reported per all-gather"""
    from seekstorm_amd import distributed as D
    sh, rows, n_docs, dim = both
    comm = D.ShardComm(0, 1, 0)
    comm.profile(True)
    nq, k = 5, 20
    qs = O.vec_gen(O.VECQ_SEED, 0, nq, dim)
    doc, score, cnt, tot = sh.search_vector_batch(qs, k)
    for _ in range(2):
        md, ms, mc, mt = comm.search_vector_sharded(sh, qs, k)
        assert np.array_equal(mc, cnt) and np.array_equal(mt, tot)
        for i in range(nq):
            assert np.array_equal(md[i, :cnt[i]], doc[i, :cnt[i]].astype(np.uint64)) and np.array_equal(ms[i, :cnt[i]], score[i, :cnt[i]])
    tl = [[3, 7, 9], [5, 2], [4], [1, 8, 6], [0, 9]]
    q = sh.make_queries(tl, S.QueryType.Union)
    ix = S.Index([sh])
    for offset, length in ((0, 15), (3, 10)):
        hd, hs, hsrc, hc, ht = comm.search_hybrid_sharded(sh, q, qs, offset, length)
        ld, ls, lc, lt = sh.search_lexical_batch(q, offset + length)
        vd, vs, vc, vt = sh.search_vector_batch(qs, offset + length)
        for i in range(nq):
            ro = ix.search(tl[i], qs[i], S.QueryType.Union, S.SearchMode.Hybrid, offset, length, normalize_query=False)
            want_ids = [r.doc_id for r in ro.results]
            want_sc = np.array([r.score for r in ro.results], np.float32)
            assert hc[i] == len(want_ids) and hd[i, :hc[i]].tolist() == want_ids
            assert np.array_equal(hs[i, :hc[i]], want_sc)  # RRF scores: the same f32 operations on the device and on the host
            assert hsrc[i, :hc[i]].tolist() == [int(r.source) for r in ro.results]
            assert int(ht[i]) == max(int(lt[i]), int(vt[i])) == ro.result_count_total
    n, us = comm.profile_read()
    assert n == 2 + 2 and 0.0 < us < 5e4  # one all-gather per sharded call
    comm.close()


def test_sharded_search_reports_a_local_failure_instead_of_hanging(S, O, both):
    """a rank whose own search fails (no image; a term its shard does not have) still enters the collective and returns ITS error
    -- with more ranks the others would return SS_EPEER (the gloo world-2 test drives that through the protocol's mirror)"""
    from seekstorm_amd import _native as N
    from seekstorm_amd import distributed as D
    sh, rows, n_docs, dim = both
    comm = D.ShardComm(0, 1, 0)
    empty = S.Shard(0)
    qs = O.vec_gen(O.VECQ_SEED, 0, 2, dim)
    with pytest.raises(N.SeekStormHipError) as e:
        comm.search_vector_sharded(empty, qs, 10)
    assert e.value.code == -5  # SS_ESTATE
    q = sh.make_queries([[3, 7], [5]], S.QueryType.Union)
    q["term"][1][0] = 0xFFFFF0  # not a term of this shard
    with pytest.raises(N.SeekStormHipError) as e:
        comm.search_lexical_sharded(sh, q, 10)
    assert e.value.code == -1
    # the communicator is still usable afterwards
    q = sh.make_queries([[3, 7], [5]], S.QueryType.Union)
    md, ms, mc, mt = comm.search_lexical_sharded(sh, q, 10)
    d, s_, c, t = sh.search_lexical_batch(q, 10)
    assert np.array_equal(mc, c) and np.array_equal(mt, t) and np.array_equal(ms, s_)
    assert N.lib().ss_strerror(-6).decode().startswith("a collective search failed")
    empty.close()
    comm.close()


def test_i8_euclidean_with_scales_needs_both_norms(S, O):
    """euclidean_i8_quantized = max(0, n1 + n2 - 2 dot s1 s2): with scales but without the record norms (ss_vec_set_row_norms) or
    the query norm the ranking would silently be wrong -- the search refuses instead (the reference always carries both norms)"""
    from seekstorm_amd import _native as N
    rows = O.quantize_i8(O.vec_gen(5, 0, 2000, 64))
    q8 = O.quantize_i8(O.vec_gen(6, 0, 2, 64))
    sh = S.Shard(0)
    sh.set_vector_similarity("euclidean")
    scale = np.full(2000, 0.5, np.float32)
    sh.upload_vectors_i8(rows, row_scale=scale)
    L = N.lib()
    doc = np.zeros((2, 5), np.uint32); sc = np.zeros((2, 5), np.float32); cnt = np.zeros(2, np.uint32); tot = np.zeros(2, np.uint64)
    qscale = np.full(2, 0.25, np.float32)
    qnorm = np.full(2, 3.0, np.float32)
    args = lambda qn: (sh._h, 2, q8.ctypes.data_as(C.c_void_p), N.ptr(qscale, N.f32p), qn, 5, N.FLT_MIN_NEG, None, N.ptr(doc, N.u32p),
                       N.ptr(sc, N.f32p), N.ptr(cnt, N.u32p), N.ptr(tot, N.u64p), None)
    assert L.ss_vec_search_i8_euclid(*args(N.ptr(qnorm, N.f32p))) == -5  # SS_ESTATE: no record norms yet
    sh.set_row_norms(np.full(2000, 2.0, np.float32))
    assert L.ss_vec_search_i8_euclid(*args(None)) == -1                  # SS_EINVAL: no query norm
    assert L.ss_vec_search_i8_euclid(*args(N.ptr(qnorm, N.f32p))) == 0
    sh.close()


def test_upload_positions_checks_the_array_length_first(S, O):
    """ss_bm25_upload_positions with fewer positions than sum(tf), or none at all: SS_EINVAL before any posting is walked (the
    walk indexes the array by the running sum of the tfs), and no image is left behind"""
    from seekstorm_amd import _native as N
    n_docs = 5000
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = O.lex_corpus(n_docs, [3000, 3600])
    tfs = np.maximum(tfs, 2).astype(np.uint16)  # every posting has >= 2 positions
    need = int(tfs.sum())
    pos = np.concatenate([np.arange(1, t + 1, dtype=np.uint16) for t in tfs])
    sh = S.Shard(0)
    L = N.lib()
    up = lambda p, n: L.ss_bm25_upload_positions(sh._h, n_docs, N.ptr(dl, N.u8p), 2, N.ptr(offs, N.u64p), N.ptr(docs, N.u32p), N.ptr(tfs, N.u16p), p, n)
    assert up(N.ptr(pos[:need // 2].copy(), N.u16p), need // 2) == -1
    assert up(None, 0) == -1
    assert up(None, need) == -1
    n, a, t, p = C.c_uint64(), C.c_float(), C.c_uint32(), C.c_uint64()
    assert L.ss_bm25_info(sh._h, C.byref(n), C.byref(a), C.byref(t), C.byref(p)) == -5  # SS_ESTATE: nothing was built
    assert up(N.ptr(pos, N.u16p), need) == 0
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


def test_all_terms_frequent_over_several_indexed_fields(S, O):
    """decode_positions_multiterm_multifield's form of the shortcut (add_result.rs:1595-1607, 3111-3122): when N > 256 k and every
    term of an intersection is in at least half of the docs, a doc is counted but ranked only if every term has >= 10 positions in the
    LOWEST field that holds the doc (an embedded pointer -- <= 4 positions -- or a record whose first field has < 10).  Over the
    image's merged lists; off under a field filter (3116); both mirrors mark the query like the reference"""
    rng = np.random.default_rng(91)
    n_docs, n_fields = 50_000, 3
    dl = np.stack([O.lex_doclen(n_docs, seed=O.LEX_SEED + 3 * f) for f in range(n_fields)])
    boost = np.array([1.5, 1.0, 0.5], np.float32)
    dfs = [34_000, 30_000, 27_000, 6_000]
    offs, D, F, T = [0], [], [], []
    for df in dfs:
        for d in np.sort(rng.choice(n_docs, df, replace=False)):
            fs = np.sort(rng.choice(n_fields, size=int(rng.integers(1, n_fields + 1)), replace=False))
            for f in fs:
                D.append(int(d)); F.append(int(f)); T.append(int(min(rng.geometric(0.2), 600)))  # ~13 % of the entries have tf >= 10
        offs.append(len(D))
    offs, D, F, T = np.array(offs, np.uint64), np.array(D, np.uint32), np.array(F, np.uint8), np.array(T, np.uint16)
    sh = S.Shard(0)
    sh.upload_lexical_fields(n_docs, dl, boost, offs, D, F, T)
    assert sh.fields_info() == (3, True, False)
    gone = list(range(1, n_docs, 61))
    sh.set_deleted(gone)
    cases = [[0, 1], [0, 1, 2], [0, 3], [1, 2]]
    flagged = [True, True, False, True]
    q = sh.make_queries(cases, S.QueryType.Intersection)
    rel = 1e-4
    differs = 0
    for k in (10, 150, 250):  # 50 000 > 256 * 150 but not > 256 * 250
        marked = sh.mark_all_terms_frequent(q, k)
        assert [bool(x >> 31) for x in marked["op"]] == [f and k < 250 for f in flagged]
        for strat in (0, 1):
            sh.set_strategy(strat)
            for rt in (S.ResultType.TopkCount, S.ResultType.Topk):
                doc, score, cnt, tot = sh.search_lexical_batch(q, k, rt)
                for i, terms in enumerate(cases):
                    plain = O.search_fields_exhaustive(n_docs, dl, boost, offs, D, F, T, terms, O.OP_AND, k, deleted=gone)
                    if flagged[i] and k < 250:
                        od, os_, otot = O.search_fields_shortcut(n_docs, dl, boost, offs, D, F, T, terms, k, deleted=gone)
                        differs += int(not np.array_equal(od, plain[0]))
                    else:
                        od, os_, otot = plain[:3]
                    if rt == S.ResultType.TopkCount:
                        assert int(tot[i]) == otot == plain[2]
                    n = int(cnt[i])
                    assert n == len(od) and np.allclose(score[i][:n], os_, rtol=rel)
                    if n:
                        band = abs(float(os_[-1])) * rel
                        clear = lambda dd, ss: {int(x) for x, y in zip(dd, ss) if y > os_[-1] + 2 * band}
                        assert clear(doc[i][:n], score[i][:n]) <= set(od.tolist()) and clear(od, os_) <= set(doc[i][:n].tolist())
    assert differs >= 6  # the shortcut really changes answers here
    sh.set_strategy(0)
    # under a field filter the reference switches the shortcut off: the marked bit changes nothing
    qf = sh.make_queries([[0, 1]], S.QueryType.Intersection, field_filter=[0, 2])
    a = sh.search_lexical_batch(qf, 10)
    qm = qf.copy()
    qm["op"][0] |= 0x80000000
    b = sh.search_lexical_batch(qm, 10, reference_shortcuts=False)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    od, os_, otot, _ = O.search_fields_exhaustive(n_docs, dl, boost, offs, D, F, T, [0, 1], O.OP_AND, 10, deleted=gone, field_filter=[0, 2])
    assert int(a[3][0]) == otot and np.allclose(a[1][0][:len(od)], os_, rtol=rel)
    sh.close()
