"""Pins the CPU oracle: formula KATs, the reference's own count assertions
(tests/test.rs:150-208, 676-744 re-enacted on hand-built postings), container round-trips,
and agreement with the independent naive numpy restatement."""
import numpy as np
import pytest

from oracle import naive
from oracle import oracle as O


def test_smallfloat_kats():
    L = O.lib()
    for i in range(40):
        assert L.so_int_to_byte4(i) == i  # 0..39 -> identity
    # SURVEY 8c derived KATs
    assert L.so_int_to_byte4(100) == 57 and L.so_byte4_to_int(57) == 96
    assert L.so_int_to_byte4(1000) == 87 and L.so_byte4_to_int(87) == 984
    assert L.so_int_to_byte4(65535) == 135 and L.so_byte4_to_int(135) == 61464
    assert L.so_byte4_to_int(100) == 3096
    assert L.so_byte4_to_int(200) == 16777240
    assert L.so_byte4_to_int(255) == 2013265944
    # round trip is monotone and idempotent; both restatements agree everywhere
    for b in range(256):
        v = L.so_byte4_to_int(b)
        assert v == naive.byte4_to_int(b)
        assert L.so_int_to_byte4(v) == b
    for i in list(range(0, 5000)) + [65535, 10 ** 6, 2 ** 31]:
        assert L.so_int_to_byte4(i) == naive.int_to_byte4(i)


def test_bm25_kat():
    L = O.lib()
    idf = L.so_idf(1_000_000, 1000)
    assert abs(idf - 6.9072566) < 1e-5
    comp = np.zeros(256, np.float32)
    L.so_bm25_component_cache(np.float32(120.5), comp.ctypes.data_as(O.f32p))
    b = L.so_int_to_byte4(100)
    assert abs(comp[b] - 1.0170125) < 1e-6
    s = L.so_bm25_term(idf, 3, float(comp[b]))
    assert abs(s - 11.348706) < 1e-4
    assert np.allclose(comp, naive.component_cache(np.float32(120.5)), rtol=1e-7)
    assert abs(float(naive.idf(1_000_000, 1000)) - idf) < 1e-6


def test_rrf_kat():
    d, s, src = O.merge(2, lex=([10, 11, 12], [3.0, 2.0, 1.0]), vec=([20, 21, 22], [0.9, 0.8, 0.7]), length=6)
    assert np.allclose(sorted(s, reverse=True)[:6:2], [1.6666666, 0.625, 0.3846154], rtol=1e-6)
    # doc in both lists -> sum, source Hybrid (search.rs:1995-2008)
    d, s, src = O.merge(2, lex=([5, 6], [2.0, 1.0]), vec=([6, 7], [0.5, 0.4]), length=3)
    assert list(d) == [6, 5, 7]
    assert np.isclose(s[0], 1.0 / 1.6 + 1.0 / 0.6, rtol=1e-6) and src[0] == 2
    nd, ns = naive.rrf([5, 6], [6, 7], 3)
    assert nd == list(d) and np.allclose(ns, s, rtol=1e-6)


def _mini_index():
    """4-doc fixture shaped like tests/test.rs:96-148: term ids 0='test', 1='body2'."""
    # docs 0..3; 'test' occurs in docs 1 and 2; 'body2' in doc 1 only
    doclen = np.array([O.lib().so_int_to_byte4(x) for x in (3, 4, 4, 3)], np.uint8)
    offs = np.array([0, 2, 3], np.uint64)
    docs = np.array([1, 2, 1], np.uint32)
    tfs = np.array([1, 1, 1], np.uint16)
    return O.Shard(4, doclen, offs, docs, tfs)


def test_reference_count_pins():
    sh = _mini_index()
    # tests/test.rs:150-177  "+body2 +test" -> results.len()==1, result_count_total==1
    d, s, tot = sh.search([1, 0], O.OP_AND, 10, O.RT_TOPKCOUNT)
    assert len(d) == 1 and tot == 1 and d[0] == 1
    # tests/test.rs:181-208  "test" union Count -> result_count_total == 2
    d, s, tot = sh.search([0], O.OP_OR, 10, O.RT_COUNT)
    assert len(d) == 0 and tot == 2
    d, s, tot = sh.search([0, 1], O.OP_OR, 10, O.RT_TOPKCOUNT)
    assert tot == 2 and set(d) == {1, 2} and d[0] == 1


def test_reference_vector_count_pin():
    # tests/test.rs:676-744: 3 vectors, k=10 -> 3 hits (TopK with len < k returns everything)
    rows = O.vec_gen(7, 0, 3, 128)
    d, s, tot, obs = O.vec_search(rows, rows[1], 10)
    assert len(d) == 3 and tot == 3 and obs == 3 and d[0] == 1 and abs(s[0] - 1.0) < 1e-5
    assert all(s[i] >= s[i + 1] for i in range(len(s) - 1))


def test_dot_generators_match_reference_tolerance():
    # vector_similarity.rs:3012-3037 deterministic generator; |simd - scalar| < 1e-3 on 128-d
    i = np.arange(128, dtype=np.float32)
    a = (np.sin(0.137 * i) / 2 + np.cos(0.013 * i) / 2).astype(np.float32)
    b = (np.sin(0.137 * (i + 7)) / 2 + np.cos(0.013 * (i + 7)) / 2).astype(np.float32)
    L = O.lib()
    s0 = L.so_dot_f32(a.ctypes.data_as(O.f32p), b.ctypes.data_as(O.f32p), 128)
    s1 = L.so_dot_f32_lanes8(a.ctypes.data_as(O.f32p), b.ctypes.data_as(O.f32p), 128)
    assert abs(s0 - s1) < 1e-3
    assert abs(s0 - float(a.astype(np.float64) @ b.astype(np.float64))) < 1e-3
    n = O.normalize(a)
    assert abs(float((n.astype(np.float64) ** 2).sum()) - 1.0) < 1e-5  # vector_similarity.rs:3103-3112


def test_vector_score_field_and_threshold():
    L = O.lib()
    assert abs(L.so_vector_score_field(1.0) - ((1.0 / 16129.0) + 1.0) / 2.0) < 1e-7
    assert abs(L.so_threshold_raw(0.7) - ((0.7 * 2 - 1) * 16129.0)) < 1e-2


def _corpus(n_docs, terms, seed=O.LEX_SEED):
    dl = O.lex_doclen(n_docs, seed)
    offs, docs, tfs = O.lex_corpus(n_docs, terms, seed)
    return dl, offs, docs, tfs


def test_container_chooser_and_roundtrip():
    n_docs = 200_000
    # df 0.05% (array), ~4% (array), ~20% (bitmap)
    terms = [0, 3000, 4095]
    dl, offs, docs, tfs = _corpus(n_docs, terms)
    # plus a hand-made run-heavy term -> RLE
    run_docs = np.concatenate([np.arange(100, 400), np.arange(70000, 70800)]).astype(np.uint32)
    offs = np.append(offs, offs[-1] + len(run_docs)).astype(np.uint64)
    docs = np.concatenate([docs, run_docs])
    tfs = np.concatenate([tfs, np.ones(len(run_docs), np.uint16)])
    sh = O.Shard(n_docs, dl, offs, docs, tfs)
    kinds = set()
    for t in range(4):
        d = docs[int(offs[t]):int(offs[t + 1])]
        bo = 0
        for blk in np.unique(d >> 16):
            ct, bid, cnt, mp = sh.container(t, bo)
            exp = (d[(d >> 16) == blk] & 0xFFFF).astype(np.uint16)
            assert bid == blk and cnt == len(exp)
            runs = 1 + int((np.diff(exp.astype(np.int64)) != 1).sum())
            thr = cnt // 2 if cnt < 4096 else 2048
            want = O.CT_RLE if (thr > 0 and runs - 1 < thr) else (O.CT_ARRAY if cnt < 4096 else O.CT_BITMAP)
            assert ct == want
            kinds.add(ct)
            assert np.array_equal(sh.decode_block(t, bo), exp)
            bo += 1
    assert kinds == {O.CT_ARRAY, O.CT_BITMAP, O.CT_RLE}


@pytest.mark.parametrize("op", [O.OP_AND, O.OP_OR])
@pytest.mark.parametrize("terms", [[4000, 3500], [4095, 4000, 3000], [2000, 4095], [1000]])
def test_oracle_vs_exhaustive_vs_naive(op, terms):
    n_docs = 150_000
    dl, offs, docs, tfs = _corpus(n_docs, terms)
    sh = O.Shard(n_docs, dl, offs, docs, tfs)
    q = list(range(len(terms)))
    k = 10
    d1, s1, tot1 = sh.search(q, op, k, O.RT_TOPKCOUNT)
    d2, s2, tot2 = sh.search_exhaustive(q, op, k)
    post = [(docs[int(offs[i]):int(offs[i + 1])], tfs[int(offs[i]):int(offs[i + 1])]) for i in q]
    ids, sc = naive.bm25_scores(n_docs, dl, post, op == O.OP_AND or len(terms) == 1)
    d3, s3 = naive.topk(ids, sc, k)
    eff_and = op == O.OP_AND or len(terms) == 1
    assert tot2 == len(ids)
    assert tot1 == tot2  # TopkCount is exact (intersection.rs:2227-2233, add_result.rs:3522-3536)
    assert len(d1) == len(d2) == len(d3) == min(k, len(ids))
    # scores agree within 1e-4 relative; sets agree outside the tie band of the k-th score
    assert np.allclose(s1, s2, rtol=1e-4) and np.allclose(s2, s3, rtol=1e-4)
    if len(s2):
        kth = s2[-1]
        strict = lambda d, s: {int(x) for x, y in zip(d, s) if y > kth * (1 + 1e-4)}
        assert strict(d1, s1) == strict(d2, s2) == strict(d3, s3)
    if eff_and:
        # bit-exact doc-id set of the conjunction: ask for everything
        dall, sall, tall = sh.search(q, O.OP_AND, len(ids) + 5, O.RT_TOPKCOUNT)
        assert set(map(int, dall)) == set(map(int, ids)) and tall == len(ids)
    # Topk (with early termination) returns the same ranking as TopkCount
    d4, s4, _ = sh.search(q, op, k, O.RT_TOPK)
    assert np.allclose(s4, s1, rtol=1e-6)


def test_vector_oracle_vs_naive():
    rows = O.vec_gen(O.VEC_SEED, 0, 5000, 64)
    qs = O.vec_gen(O.VECQ_SEED, 0, 4, 64)
    for q in qs:
        d, s, tot, obs = O.vec_search(rows, q, 100)
        nd, ns = naive.cosine_topk(rows, q, 100)
        assert obs == 5000 and len(d) == 100
        assert np.allclose(s, ns, rtol=1e-4, atol=1e-6)
        kth = ns[-1]
        assert {int(x) for x, y in zip(d, s) if y > kth + 1e-5} == {int(x) for x, y in zip(nd, ns) if y > kth + 1e-5}


def test_vector_dedup_semantics():
    # vector.rs:441-452,462-473: several records of one doc -> one result with the max score
    rows = O.vec_gen(3, 0, 50, 32)
    docs = (np.arange(50) // 2).astype(np.uint32)
    d, s, tot, obs = O.vec_search(rows, rows[10], 10, row_doc_ids=docs)
    assert len(set(map(int, d))) == len(d) and d[0] == 5
    full = rows @ rows[10]
    for doc, sc in zip(d, s):
        assert abs(sc - full[docs == doc].max()) < 1e-5


def test_generator_is_stable():
    # golden values of the counter-based generator (guards CPU<->GPU agreement of the hash)
    L = O.lib()
    assert L.so_splitmix64(0) == 0xE220A8397B1DCDAF
    assert L.so_h(O.LEX_SEED, 1, 2) == L.so_h(O.LEX_SEED, 1, 2)
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "generator.npz"))
    d, t = O.lex_term(4000, O.term_thresholds()[4000], 50_000)
    assert np.array_equal(d[:64], g["lex_docs"]) and np.array_equal(t[:64], g["lex_tfs"])
    assert np.array_equal(O.lex_doclen(4096)[:256], g["doclen"])
    assert np.array_equal(O.vec_gen(O.VEC_SEED, 5, 2, 16), g["vec"])


def test_deleted_docs_neither_count_nor_rank():
    # delete_hashset: add_result.rs:3435 (skipped before counting), union.rs:975 (cleared from the count bitmaps),
    # vector.rs:1450-1452 (scored, not pushed)
    n_docs = 150_000
    terms = [4095, 4000, 3000]
    dl, offs, docs, tfs = _corpus(n_docs, terms)
    sh = O.Shard(n_docs, dl, offs, docs, tfs)
    q = [0, 1, 2]
    for op in (O.OP_OR, O.OP_AND):
        d0, s0, tot0 = sh.search(q, op, 10, O.RT_TOPKCOUNT)
        gone = [int(d0[0]), int(d0[3]), 7]  # two ranked docs + one arbitrary doc
        member = lambda d: sum(d in docs[int(offs[i]):int(offs[i + 1])] for i in q)
        matched_gone = sum((member(g) == 3) if op == O.OP_AND else (member(g) > 0) for g in gone)
        sh.set_deleted(gone)
        for fn in (lambda: sh.search(q, op, 10, O.RT_TOPKCOUNT), lambda: sh.search_exhaustive(q, op, 10)):
            d1, s1, tot1 = fn()
            assert tot1 == tot0 - matched_gone
            assert not set(map(int, d1)) & set(gone)
            keep = [i for i, d in enumerate(d0) if int(d) not in gone]
            assert np.allclose(s1[:len(keep)], s0[keep], rtol=1e-6)  # the survivors move up, scores unchanged (idf keeps N, df)
        assert sh.search(q, op, 0, O.RT_COUNT)[2] == tot0 - matched_gone
        sh.set_deleted([])
        assert sh.search(q, op, 10, O.RT_TOPKCOUNT)[2] == tot0
    rows = O.vec_gen(O.VEC_SEED, 0, 3000, 64)
    qv = O.vec_gen(O.VECQ_SEED, 0, 1, 64)[0]
    d0, s0, tot0, obs0 = O.vec_search(rows, qv, 20)
    d1, s1, tot1, obs1 = O.vec_search(rows, qv, 20, deleted=[int(d0[0]), int(d0[5])])
    # observed_vector_count counts TopK::push calls (vector.rs:421): the two tombstoned records never reach it (1450-1452)
    assert obs0 == 3000 and obs1 == 2998 and int(d0[0]) not in d1 and int(d0[5]) not in d1
    assert [int(x) for x in d1[:4]] == [int(x) for x in d0[1:5]]


def test_not_terms_exclude_docs():
    # not_query_list (add_result.rs:3440-3497; union.rs:483-530): set difference, applied before counting
    n_docs = 150_000
    terms = [4095, 4000, 3000, 3500]
    dl, offs, docs, tfs = _corpus(n_docs, terms)
    sh = O.Shard(n_docs, dl, offs, docs, tfs)
    lists = [set(map(int, docs[int(offs[i]):int(offs[i + 1])])) for i in range(4)]
    for op, pos, neg in ((O.OP_OR, [0, 1], [2]), (O.OP_AND, [0, 1], [3]), (O.OP_OR, [1], [0, 2]), (O.OP_AND, [0, 2], [1, 3])):
        base = set.intersection(*[lists[i] for i in pos]) if op == O.OP_AND else set.union(*[lists[i] for i in pos])
        want = base - set.union(*[lists[i] for i in neg])
        d1, s1, t1 = sh.search(pos, op, 10, O.RT_TOPKCOUNT, not_terms=neg)
        d2, s2, t2 = sh.search_exhaustive(pos, op, 10, not_terms=neg)
        assert t1 == t2 == len(want)
        assert set(map(int, d1)) <= want and set(map(int, d2)) <= want
        assert np.allclose(s1, s2, rtol=1e-4)
        # scores of the survivors are those of the query without NOT terms
        dall, sall, _ = sh.search_exhaustive(pos, op, 2000)
        keep = [(int(d), float(s)) for d, s in zip(dall, sall) if int(d) in want][:10]
        assert [d for d, _ in keep] == [int(d) for d in d2][:len(keep)]


def test_i8_quantisation_and_integer_dot():
    # quantize_f32_to_i8 (vector_similarity.rs:1226-1232): round half away from zero, clamp +-127; dot_i8 exact (1011-1016)
    v = np.array([0.5 / 127, 1.5 / 127, -0.5 / 127, -1.5 / 127, 2.0, -3.0, 0.3, 0.0, 126.4 / 127], np.float32)
    assert O.quantize_i8(v).tolist() == [1, 2, -1, -2, 127, -127, 38, 0, 126]
    rows = O.quantize_i8(O.vec_gen(O.VEC_SEED, 0, 400, 96))
    q = O.quantize_i8(O.vec_gen(O.VECQ_SEED, 0, 1, 96))[0]
    d, s, tot, obs = O.vec_search_i8(rows, q, 10)
    full = rows.astype(np.int64) @ q.astype(np.int64)
    order = np.lexsort((np.arange(400), -full))[:10]
    assert obs == 400 and [int(x) for x in s] == [int(full[i]) for i in order]
    assert {int(x) for x, y in zip(d, s) if y > s[-1]} == {int(i) for i in order if full[i] > s[-1]}
    # dot_i8_quantized (1754-1758): dot as f32 * query_scale * embedding_scale, in that order
    rs = np.linspace(0.5, 2.0, 400).astype(np.float32)
    d2, s2, _, _ = O.vec_search_i8(rows, q, 10, row_scale=rs, query_scale=0.37)
    want = (full.astype(np.float32) * np.float32(0.37)) * rs
    assert np.array_equal(np.sort(s2)[::-1], s2) and np.allclose(s2, np.sort(want)[::-1][:10], rtol=0, atol=0)


def test_ann_modes_select_clusters_like_the_reference():
    # vector.rs:1300-1392: per level, TopK::new(min(n_probe, clusters), cluster threshold) over the medoids (first record
    # of each cluster), survivors sorted by score desc, their records visited cluster after cluster
    rng = np.random.default_rng(3)
    dim, lc = 16, [3, 2]
    child = np.array([4, 6, 5, 7, 3], np.uint32)
    rows = rng.standard_normal((25, dim)).astype(np.float32)
    q = rng.standard_normal(dim).astype(np.float32)
    first = np.concatenate([[0], np.cumsum(child.astype(np.int64))[:-1]])
    med = np.array([O.vec_search(rows[f:f + 1], q, 1)[1][0] for f in first])  # medoid scores by the same dot
    # Nprobe(1): the best medoid of each level
    best = [int(np.argmax(med[:3])), 3 + int(np.argmax(med[3:]))]
    visited = np.concatenate([np.arange(first[c], first[c] + child[c]) for c in best])
    d, s, tot, obs, ncl = O.vec_search_ann(rows, q, 25, lc, child, n_probe=1)
    assert ncl == 2 and obs == len(visited) and set(map(int, d)) == set(map(int, visited))
    assert np.all(s[:-1] >= s[1:])
    # every cluster: AnnMode::All's result
    d_all, s_all, _, _ = O.vec_search(rows, q, 10)
    d2, s2, _, obs2, ncl2 = O.vec_search_ann(rows, q, 10, lc, child)
    assert ncl2 == 5 and obs2 == 25 and np.array_equal(s2, s_all) and set(map(int, d2)) == set(map(int, d_all))
    # Similaritythreshold: clusters whose medoid scores below the threshold are skipped (score < threshold -> reject)
    thr = float(np.sort(med)[2])
    d3, s3, _, obs3, ncl3 = O.vec_search_ann(rows, q, 25, lc, child, cluster_threshold_raw=thr)
    keep = [c for c in range(5) if med[c] >= thr]
    assert ncl3 == len(keep) == 3 and obs3 == int(child[keep].sum())
    # ties between medoids keep the EARLIER cluster (TopK::push admits only score > current minimum)
    rows_t = rows.copy()
    rows_t[first[1]] = rows_t[first[0]]
    rows_t[first[2]] = rows_t[first[0]]
    d4, _, _, obs4, _ = O.vec_search_ann(rows_t[:15], q, 15, [3], child[:3], n_probe=1)
    assert obs4 == child[0] and set(map(int, d4)) == set(range(int(child[0])))
    # i8 records: same walk over exact integer scores
    r8, q8 = O.quantize_i8(rows / np.abs(rows).max()), O.quantize_i8(q / np.abs(q).max())
    med8 = r8[first].astype(np.int64) @ q8.astype(np.int64)
    d5, s5, _, obs5, ncl5 = O.vec_search_i8_ann(r8, q8, 25, lc, child, n_probe=1)
    best8 = [int(np.argmax(med8[:3])), 3 + int(np.argmax(med8[3:]))]
    assert ncl5 == 2 and obs5 == int(child[best8].sum())
    assert sorted(map(int, s5), reverse=True) == [int(x) for x in s5]


def test_all_terms_frequent_condition_and_rule():
    # intersection.rs:198-209: indexed_doc_count > top_k << 8 (strict) and posting_count / indexed_doc_count >= 0.5 for every
    # term; under it a doc with some tf < 10 (embedded pointers hold <= 4 positions) is counted but not ranked
    n_docs = 5120
    dl = np.full(n_docs, 40, np.uint8)
    a = np.arange(0, n_docs, 2, dtype=np.uint32)            # df = N / 2 exactly: frequent
    b = np.arange(0, n_docs, 2, dtype=np.uint32)[:-1]       # df = N / 2 - 1: not frequent
    c = np.arange(0, n_docs, dtype=np.uint32)               # every doc
    tf_a = np.where(a % 8 == 0, 12, 3).astype(np.uint16)
    tf_c = np.where(c % 16 == 0, 30, 9).astype(np.uint16)
    offs = np.array([0, len(a), len(a) + len(b), len(a) + len(b) + len(c)], np.uint64)
    sh = O.Shard(n_docs, dl, offs, np.concatenate([a, b, c]), np.concatenate([tf_a, np.ones(len(b), np.uint16), tf_c]))
    assert sh.all_terms_frequent([0, 2], 19) and not sh.all_terms_frequent([0, 2], 20)   # 5120 > 19 * 256 = 4864, not > 5120
    assert not sh.all_terms_frequent([1, 2], 10)
    plain = sh.search_exhaustive([0, 2], O.OP_AND, 10)
    short = sh.search_exhaustive([0, 2], O.OP_AND, 10, reference_shortcuts=True)
    assert plain[2] == short[2] == len(a)                    # every match is counted either way
    rankable = {int(d) for d in a if d % 16 == 0}            # tf 12 and 30: both >= 10
    assert set(map(int, short[0])) <= rankable and len(short[0]) == 10
    few = sh.search_exhaustive([0, 2], O.OP_AND, 19, reference_shortcuts=True)
    assert len(few[0]) == 19
    # an intersection that is not all-frequent, a union, a single term: untouched
    for terms, op in (([1, 2], O.OP_AND), ([0, 2], O.OP_OR), ([0], O.OP_AND)):
        x, y = sh.search_exhaustive(terms, op, 10), sh.search_exhaustive(terms, op, 10, reference_shortcuts=True)
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])


def test_field_filter_rule_on_bm25f():
    # add_result.rs:3124-3136: every term must occur in a listed field; the score still sums all fields
    n_docs = 6
    dl = np.full((2, n_docs), 20, np.uint8)
    # term 0: doc0 field0, doc1 field1, doc2 fields 0+1; term 1: doc0 field1, doc2 field0, doc3 field0
    offs = np.array([0, 4, 7], np.uint64)
    docs = np.array([0, 1, 2, 2, 0, 2, 3], np.uint32)
    fields = np.array([0, 1, 0, 1, 1, 0, 0], np.uint8)
    tfs = np.array([1, 2, 1, 3, 1, 2, 5], np.uint16)
    args = (n_docs, dl, None, offs, docs, fields, tfs)
    assert O.search_fields_exhaustive(*args, [0, 1], O.OP_AND, 10)[2] == 2                            # docs 0, 2
    d, s, tot, _ = O.search_fields_exhaustive(*args, [0, 1], O.OP_AND, 10, field_filter=[0])
    assert tot == 1 and list(d) == [2]                                                               # doc 0 has term 1 only in field 1
    full = O.search_fields_exhaustive(*args, [0, 1], O.OP_AND, 10)
    assert s[0] == full[1][list(full[0]).index(2)]                                                   # score over all fields
    assert O.search_fields_exhaustive(*args, [0], O.OP_OR, 10, field_filter=[1])[2] == 2             # docs 1, 2


# ------------------------------------------------------------------ reference-structured dispatch (round 2)
def test_reference_structured_dispatch_equals_table_scan():
    """so_search_lex_ref -- single_blockid / union_docid_2 / union_docid_3 with the sub-query queue and add_topk's
    docid_hashset arm (single.rs:292-417, union.rs:1168-1479, min_heap.rs:1193-1260) -- returns what the union_scan /
    intersection restatement returns: scores, counts, every result type, NOT terms, tombstones.  The one place they may
    differ is a quirk of the reference kept on purpose: a 2-term union's count under a delete set without NOT terms
    (union.rs:1240-1249: posting_count sums minus the intersection count) still counts the deleted docs."""
    from oracle import oracle as O
    n_docs = 200_000
    voc = list(range(2000, 4096, 60))
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = O.lex_corpus(n_docs, voc)
    sh = O.Shard(n_docs, dl, offs, docs, tfs)
    rng = np.random.default_rng(5)
    quirk = 0
    for it in range(300):
        nt = int(rng.integers(1, 7))
        terms = [int(x) for x in rng.choice(len(voc), nt, replace=False)]
        op = O.OP_OR if rng.random() < 0.7 else O.OP_AND
        k = int(rng.choice([1, 10, 100]))
        rt = int(rng.choice([O.RT_TOPK, O.RT_TOPKCOUNT, O.RT_COUNT]))
        nots = [int(x) for x in rng.choice(len(voc), int(rng.integers(0, 2)), replace=False) if int(x) not in terms]
        deleted = it % 5 == 0
        if deleted:
            sh.set_deleted(rng.choice(n_docs, 1500, replace=False))
        elif it % 5 == 1:
            sh.set_deleted([])
            deleted = False
        a = sh.search(terms, op, k, rt, nots)
        b = sh.search_ref(terms, op, k, rt, nots)
        assert len(a[1]) == len(b[1]) and np.allclose(a[1], b[1], rtol=1e-6), (terms, op, k, rt, nots)
        if rt != O.RT_TOPK and a[2] != b[2]:
            assert op == O.OP_OR and nt == 2 and not nots, (terms, op, rt, a[2], b[2])  # only the documented quirk
            quirk += 1
    sh.set_deleted([])
    # the decomposition prunes: a 3-term union over long lists is answered from far fewer postings than the table scan reads;
    # here only equality matters, and that k docs come back sorted
    d, s, t = sh.search_ref([0, 1, 2], O.OP_OR, 10, O.RT_TOPK)
    assert len(d) == 10 and np.all(s[:-1] >= s[1:])


def test_split_corpus_and_cpu_baseline_harness():
    """split_corpus = the reference's document partitioning (doc g -> shard g % S, local g // S, index.rs:5284); the harness
    (so_bench_lex: S shard tasks per query + merge, throughput and latency modes) answers and merges what the single shard
    answers when the statistics are shard-local"""
    from oracle import oracle as O
    n_docs, S = 120_001, 4
    voc = [3000, 3400, 3800, 4000]
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = O.lex_corpus(n_docs, voc)
    parts = O.split_corpus(n_docs, dl, offs, docs, tfs, S)
    assert sum(p[0] for p in parts) == n_docs
    for sh, (nd, d, o, dd, tt) in enumerate(parts):
        assert len(d) == nd and np.array_equal(d, dl[sh::S])
        for t in range(len(voc)):
            ref = docs[int(offs[t]):int(offs[t + 1])]
            m = ref % S == sh
            assert np.array_equal(dd[int(o[t]):int(o[t + 1])].astype(np.int64) * S + sh, ref[m])
            assert np.array_equal(tt[int(o[t]):int(o[t + 1])], tfs[int(offs[t]):int(offs[t + 1])][m])
    shards = [O.Shard(*p) for p in parts]
    qs = np.array([[0, 1, 2], [1, 2, 3]], np.uint32)
    for mode, threads in ((0, 2), (1, S)):
        qps, done, lat = O.bench_lex(shards, qs, O.OP_OR, 10, O.RT_TOPK, mode, threads, 0.2)
        assert qps > 0 and done > 0 and (mode == 0 or len(lat) == done)


def test_phrase_match_reference_loop_equals_definition():
    """the reference's phrase merge loop (add_result.rs:3596-3684, restated) decides what the definition decides -- some start
    carries word i at start + i -- over random position lists, repeated words, 2..6 words"""
    from oracle import oracle as O
    rng = np.random.default_rng(17)
    hits = 0
    for it in range(4000):
        n = int(rng.integers(2, 7))
        span = int(rng.integers(6, 60))
        uniq = [np.sort(rng.choice(span, int(rng.integers(1, min(span, 12))), replace=False)) for _ in range(n)]
        seq = [int(rng.integers(0, n)) for _ in range(n)] if it % 3 == 0 else list(range(n))  # repeated words share a list
        lists = [uniq[i] for i in seq]
        if it % 5 == 0:  # plant a match
            st = int(rng.integers(0, span))
            lists = [np.unique(np.append(l, st + i)) for i, l in enumerate(lists)]
            if len(set(seq)) < n:  # shared lists must stay shared
                continue
        a, b = O.phrase_match(lists, True), O.phrase_match(lists, False)
        assert a == b, (lists,)
        hits += a
    assert 200 < hits < 3800


def test_phrase_entries_with_places_ngram_keys():
    """n-gram keys in a phrase (search.rs:3305-3328): an entry's place = entries before it + the extra places of the keys before it.
    (a) the restated merge loop agrees with the definition for arbitrary places; (b) on a corpus laid out like the reference's
    default index, a phrase whose entries are keys matches exactly the docs the phrase over the single terms matches, and a
    one-entry phrase is a term query over the key."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import ngram_corpus as NG
    from test_gpu_phrase import _corpus
    rng = np.random.default_rng(23)
    hits = 0
    for it in range(3000):
        n = int(rng.integers(2, 6))
        span = int(rng.integers(8, 60))
        lists = [np.sort(rng.choice(span, int(rng.integers(1, min(span, 10))), replace=False)) for _ in range(n)]
        places = np.cumsum([0] + [int(rng.integers(1, 4)) for _ in range(n - 1)])  # entries span 1..3 places
        if it % 4 == 0:
            st = int(rng.integers(0, span))
            lists = [np.unique(np.append(l, st + int(pl))) for l, pl in zip(lists, places)]
        a, b = O.phrase_match(lists, True, places), O.phrase_match(lists, False, places)
        assert a == b, (lists, places)
        hits += a
    assert 300 < hits < 2800
    n_docs = 20_000
    C = NG.build(O, _corpus, n_docs, [6_000, 3_000, 4_000, 900], [([0, 1], 120), ([0, 1, 2], 90), ([3, 0, 1], 40), ([0, 1, 3], 50), ([0, 1, 2, 3], 30),
                                                                  ([2, 0, 1], 40), ([0, 1, 0, 1], 20), ([3, 0, 1, 2], 25)], 12)
    osh = C.oracle_shard(O)
    single = O.Shard(n_docs, C.dl, C.offs, C.docs, C.tfs)
    single.set_positions(C.positions)
    for ph, same in zip(NG.PHRASES, NG.SAME_DOCS_AS):
        uniq, seq, places, idf = C.oracle_query(ph, lambda e, c: 1.0 + c, lambda l: 2.0)
        od, os_, tot = osh.search_phrase_items(uniq, seq, places, 20_000, idf=idf)
        od2, _, tot2 = osh.search_phrase_items(uniq, seq, places, 20_000, idf=idf, reference_loop=False)
        su = list(dict.fromkeys(same))
        sd, _, stot = single.search_phrase(su, [su.index(w) for w in same], 20_000)
        assert tot == tot2 == stot > 0 and set(od.tolist()) == set(od2.tolist()) == set(sd.tolist()), (ph, tot, tot2, stot)
    uniq, seq, places, idf = C.oracle_query([NG.AB], lambda e, c: 1.0, lambda l: 1.0)
    assert osh.search_phrase_items(uniq, seq, places, 10, idf=idf)[2] == len(C.rows_of[NG.AB])


def test_phrase_over_several_fields_rules():
    """so_search_fields_phrase (add_result.rs:2964-3414): the phrase must stand inside ONE field; a field filter lists the fields it may
    stand in; the score sums ALL fields of the unique terms; with one field it is so_search_phrase"""
    n_docs = 8
    dl = np.full((2, n_docs), 20, np.uint8)
    # term 0 ("a"), term 1 ("b"); entries sorted by (doc, field)
    #   doc 0: a@f0{3}      b@f0{4}         -> phrase in field 0
    #   doc 1: a@f0{3}      b@f1{4}         -> consecutive numbers in DIFFERENT fields: no phrase
    #   doc 2: a@f1{7, 9}   b@f1{10}        -> phrase in field 1 (9, 10)
    #   doc 3: a@f0{1} a@f1{5}   b@f0{9} b@f1{6}  -> field 0 no, field 1 yes
    #   doc 4: a@f0{65535}  b@f1{0}         -> the last position of field 0 and the first of field 1: no phrase
    #   doc 5: a@f0{2}                      -> no b
    a_ent = [(0, 0, [3]), (1, 0, [3]), (2, 1, [7, 9]), (3, 0, [1]), (3, 1, [5]), (4, 0, [65535]), (5, 0, [2])]
    b_ent = [(0, 0, [4]), (1, 1, [4]), (2, 1, [10]), (3, 0, [9]), (3, 1, [6]), (4, 1, [0])]
    ent = a_ent + b_ent
    offs = np.array([0, len(a_ent), len(ent)], np.uint64)
    docs = np.array([e[0] for e in ent], np.uint32)
    fields = np.array([e[1] for e in ent], np.uint8)
    tfs = np.array([len(e[2]) for e in ent], np.uint16)
    pos = np.array([x for e in ent for x in e[2]], np.uint16)
    for loop in (True, False):
        f = lambda **kw: O.search_fields_phrase(n_docs, dl, None, offs, docs, fields, tfs, pos, [0, 1], [0, 1], 10, reference_loop=loop, **kw)
        od, os_, tot = f()
        assert tot == 3 and sorted(od.tolist()) == [0, 2, 3]
        assert f(field_filter=(0,))[2] == 1 and f(field_filter=(0,))[0].tolist() == [0]
        assert sorted(f(field_filter=(1,))[0].tolist()) == [2, 3]
        assert f(deleted=[2])[2] == 2
        # "b a": doc 3 field 1 has b@6 ... a@5: no; nothing matches
        assert O.search_fields_phrase(n_docs, dl, None, offs, docs, fields, tfs, pos, [1, 0], [0, 1], 10, reference_loop=loop)[2] == 0
        # the score sums all fields: doc 3's equals the BM25F of the intersection {a, b}
        full = O.search_fields_exhaustive(n_docs, dl, None, offs, docs, fields, tfs, [0, 1], O.OP_AND, 10)
        for d, sc in zip(od, os_):
            assert sc == full[1][full[0].tolist().index(int(d))]
        # a filter does not change the score of a doc that still matches
        od1, os1, _ = f(field_filter=(1,))
        assert os1[od1.tolist().index(3)] == os_[od.tolist().index(3)]
    # one indexed field: the single-field phrase search, bit for bit
    rng = np.random.default_rng(9)
    n_docs, nt = 3000, 4
    dl1 = O.lex_doclen(n_docs)
    offs1, d1, t1, p1 = [0], [], [], []
    for t in range(nt):
        for d in np.sort(rng.choice(n_docs, 900, replace=False)):
            ps = np.sort(rng.choice(12, int(rng.integers(1, 5)), replace=False))
            d1.append(int(d)); t1.append(len(ps)); p1 += ps.tolist()
        offs1.append(len(d1))
    sh = O.Shard(n_docs, dl1, np.array(offs1, np.uint64), np.array(d1, np.uint32), np.array(t1, np.uint16))
    sh.set_positions(np.array(p1, np.uint16))
    for ph in ([0, 1], [2, 3, 0], [1, 1], [3, 2, 1, 0]):
        uniq = list(dict.fromkeys(ph))
        seq = [uniq.index(w) for w in ph]
        x = sh.search_phrase(uniq, seq, 20)
        y = O.search_fields_phrase(n_docs, dl1[None, :], None, offs1, d1, np.zeros(len(d1), np.uint8), t1, p1, uniq, seq, 20)
        assert x[2] == y[2] and np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] > 0


def test_all_terms_frequent_rule_over_several_fields():
    """so_search_fields_shortcut (add_result.rs:1595-1607): every matching doc is counted; a doc is ranked only if EVERY term has
    >= 10 positions in the LOWEST field that holds the doc for that term -- positions in later fields do not help"""
    n_docs = 6
    dl = np.full((3, n_docs), 30, np.uint8)
    # term 0 / term 1 entries (doc, field, tf), sorted by (doc, field)
    #   doc 0: t0 f0:12          t1 f1:10          -> ranked (both first fields >= 10)
    #   doc 1: t0 f0:9  f1:40    t1 f0:15          -> counted, not ranked (t0's lowest field has 9)
    #   doc 2: t0 f1:10          t1 f0:3 f2:50     -> counted, not ranked (t1's lowest field has 3)
    #   doc 3: t0 f2:11          t1 f2:10          -> ranked
    #   doc 4: t0 f0:20                            -> no t1: not a match
    e0 = [(0, 0, 12), (1, 0, 9), (1, 1, 40), (2, 1, 10), (3, 2, 11), (4, 0, 20)]
    e1 = [(0, 1, 10), (1, 0, 15), (2, 0, 3), (2, 2, 50), (3, 2, 10)]
    ent = e0 + e1
    offs = np.array([0, len(e0), len(ent)], np.uint64)
    docs = np.array([e[0] for e in ent], np.uint32)
    fields = np.array([e[1] for e in ent], np.uint8)
    tfs = np.array([e[2] for e in ent], np.uint16)
    od, os_, tot = O.search_fields_shortcut(n_docs, dl, None, offs, docs, fields, tfs, [0, 1], 10)
    full = O.search_fields_exhaustive(n_docs, dl, None, offs, docs, fields, tfs, [0, 1], O.OP_AND, 10)
    assert tot == full[2] == 4 and sorted(od.tolist()) == [0, 3] and sorted(full[0].tolist()) == [0, 1, 2, 3]
    for d, sc in zip(od, os_):  # a ranked doc keeps its full BM25F score
        assert sc == full[1][full[0].tolist().index(int(d))]
    assert O.search_fields_shortcut(n_docs, dl, None, offs, docs, fields, tfs, [0, 1], 10, deleted=[3])[2] == 3


def test_geo_morton_and_distances():
    """Point facets (geo_search.rs): the reference's tests hold no vectors for these, so the oracle's restatement is checked
    against what defines it -- latitude in the even bits and longitude in the odd ones of (deg * 1e7) as i32, the cast's
    truncation and saturation, decode o encode = truncation to 1e-7 deg, and the two distance formulas against numpy."""
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    lat, lon = rng.random(2000) * 180 - 90, rng.random(2000) * 360 - 180
    codes = O.morton_encode(lat, lon)
    xi = np.trunc(lat * 1e7).astype(np.int64).astype(np.int32).view(np.uint32)
    yi = np.trunc(lon * 1e7).astype(np.int64).astype(np.int32).view(np.uint32)
    for c, x, y in zip(codes[:300], xi[:300], yi[:300]):
        even = sum(((int(c) >> (2 * b)) & 1) << b for b in range(32))
        odd = sum(((int(c) >> (2 * b + 1)) & 1) << b for b in range(32))
        assert even == int(x) and odd == int(y)
    back = O.morton_decode(codes)
    assert np.array_equal(back[:, 0], xi.view(np.int32) / 1e7) and np.array_equal(back[:, 1], yi.view(np.int32) / 1e7)
    assert np.abs(back[:, 0] - lat).max() < 1.0001e-7 and np.abs(back[:, 1] - lon).max() < 1.0001e-7
    assert int(O.morton_encode([1000.0], [float("nan")])[0]) == sum(1 << (2 * b) for b in range(31))   # lat saturates to i32::MAX, NaN -> 0
    assert int(O.morton_encode([38.8951], [-77.0364])[0]) == 0xA31D06765CE7D940
    base = (38.8951, -77.0364)
    d2r, r_km = 0.017453292519943295, 6371.0087714
    x = d2r * (back[:, 1] - base[1]) * np.cos(d2r * (base[0] + back[:, 0]) / 2.0)
    y = d2r * (back[:, 0] - base[0])
    assert np.allclose(O.geo_distances(codes, base, "km"), r_km * np.sqrt(x * x + y * y), rtol=1e-12)
    assert np.allclose(O.geo_distances(codes, base, "miles") / O.geo_distances(codes, base, "km")[None, :], 3958.761315801475 / r_km, rtol=1e-12)
    xs = (base[1] - back[:, 1]) * np.cos(d2r * (back[:, 0] + base[0]) / 2.0)
    assert np.allclose(O.geo_distances(codes, base, "sortkey"), xs * xs + (base[0] - back[:, 0]) ** 2, rtol=1e-12)
    # Washington - New York: 328 km by the equirectangular formula
    ny = O.morton_encode([40.7128], [-74.0060])
    assert abs(O.geo_distances(ny, base, "km")[0] - 328.0) < 2.0
    m0, m1 = O.geo_morton_range(base, 100.0, "km")
    lat_d, lon_d = 100.0 / (d2r * r_km), 100.0 / (d2r * r_km * np.cos(d2r * base[0]))
    assert m0 == int(O.morton_encode([base[0] - lat_d], [base[1] - lon_d])[0]) and m1 == int(O.morton_encode([base[0] + lat_d], [base[1] + lon_d])[0])
