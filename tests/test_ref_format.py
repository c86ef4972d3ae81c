"""Reader of the reference's in-RAM block format (ss_ref_decode_block / ss_bm25_upload_ref_blocks) against byte arrays
written by oracle/ref_format.py (a restatement of the reference's indexing-side writers)."""
import ctypes as C

import numpy as np
import pytest

import seekstorm_amd as S
from seekstorm_amd import _native as N
from oracle import ref_format as RF


def _decode(block):
    bid, ctp, cnt, pivot, body = block
    buf = np.frombuffer(body, np.uint8).copy()
    rb = N.RefBlock(bid, ctp, cnt - 1, pivot, buf.ctypes.data, len(buf))
    d = np.zeros(65536, np.uint16)
    t = np.zeros(65536, np.uint16)
    n = N.lib().ss_ref_decode_block(C.byref(rb), N.ptr(d, N.u16p), N.ptr(t, N.u16p))
    return n, d[:max(n, 0)].copy(), t[:max(n, 0)].copy()


def _decode_positions(block):
    bid, ctp, cnt, pivot, body = block
    buf = np.frombuffer(body, np.uint8).copy()
    rb = N.RefBlock(bid, ctp, cnt - 1, pivot, buf.ctypes.data, len(buf))
    d = np.zeros(65536, np.uint16)
    t = np.zeros(65536, np.uint16)
    npos = C.c_uint64()
    n = N.lib().ss_ref_decode_block_positions(C.byref(rb), N.ptr(d, N.u16p), N.ptr(t, N.u16p), None, 0, C.byref(npos))
    assert n == -1 or npos.value == 0  # SS_EINVAL + the needed size, unless there is nothing to write
    pos = np.zeros(max(npos.value, 1), np.uint16)
    n = N.lib().ss_ref_decode_block_positions(C.byref(rb), N.ptr(d, N.u16p), N.ptr(t, N.u16p), N.ptr(pos, N.u16p), len(pos), C.byref(npos))
    return n, d[:max(n, 0)].copy(), t[:max(n, 0)].copy(), pos[:npos.value].copy()


def _case(rng, n, span, tf_hi, dense_runs=False):
    if dense_runs:
        start = int(rng.integers(0, 65536 - n))
        docs = np.arange(start, start + n)
        docs = np.delete(docs, rng.choice(n, size=max(1, n // 50), replace=False)) if n > 50 else docs
    else:
        docs = np.sort(rng.choice(span, size=n, replace=False))
    tfs = rng.integers(1, tf_hi + 1, size=len(docs))
    return docs, tfs


@pytest.mark.parametrize("n,span,tf_hi,runs,ctype", [
    (1, 65536, 1, False, RF.CT_ARRAY),
    (37, 65536, 3, False, RF.CT_ARRAY),
    (3000, 65536, 6, False, RF.CT_ARRAY),
    (5000, 65536, 4, False, RF.CT_BITMAP),
    (65536, 65536, 2, False, RF.CT_RLE),
    (4000, 65536, 9, True, RF.CT_RLE),
    (300, 65536, 700, False, RF.CT_ARRAY),
])
def test_decode_roundtrip(n, span, tf_hi, runs, ctype):
    rng = np.random.default_rng(n * 7 + tf_hi)
    docs, tfs = _case(rng, n, span, tf_hi, runs)
    blocks = RF.encode_term(docs, tfs, rng, base_bytes=bytes(rng.integers(0, 256, size=123, dtype=np.uint8)))
    assert len(blocks) == 1 and blocks[0][1] >> 30 == ctype
    cnt, d, t = _decode(blocks[0])
    assert cnt == len(docs)
    assert np.array_equal(d, docs) and np.array_equal(t, tfs)


def test_three_byte_pointers_real_limit():
    # > 32 768 bytes of position records -> pointer_pivot_p_docid inside the list (index_posting.rs:579-587)
    rng = np.random.default_rng(5)
    docs, tfs = _case(rng, 4000, 65536, 1, False)
    tfs = rng.integers(8, 30, size=len(docs))
    blk = RF.encode_term(docs, tfs, rng)[0]
    assert 0 < blk[3] < blk[2], "pivot must fall inside the list"
    cnt, d, t = _decode(blk)
    assert cnt == len(docs) and np.array_equal(d, docs) and np.array_equal(t, tfs)


def test_three_byte_embedded_forms():
    # lowered limit: ranks beyond the pivot use 3-byte pointers with 1..4 embedded positions and VINT records
    rng = np.random.default_rng(6)
    docs, tfs = _case(rng, 600, 65536, 1, False)
    tfs = rng.integers(1, 8, size=len(docs))
    blk = RF.encode_term(docs, tfs, rng, positions_limit=64, max_gap=20)[0]
    assert 0 < blk[3] < 100
    cnt, d, t = _decode(blk)
    assert cnt == len(docs) and np.array_equal(d, docs) and np.array_equal(t, tfs)


def test_embedded_pointer_bit_patterns():
    # index_posting.rs:621-640: 2-byte 10|14 bits, 11|7|7; 3-byte 1 00|21, 1 01|10|11, 1 10|7|7|7, 1 11|5|5|5|6
    assert RF.embed([0x2ABC], 2) == bytes([0xBC, 0x80 | 0x2A])
    assert RF.embed([0x55, 0x2A], 2) == bytes([(0x55 << 7 | 0x2A) & 0xFF, 0xC0 | ((0x55 << 7 | 0x2A) >> 8)])
    assert RF.embed([1, 2, 3, 4], 3)[2] >> 5 == 0b111
    assert RF.embed([5], 3)[2] >> 5 == 0b100
    assert not RF.embeddable([0x4000], 2) and RF.embeddable([0x3FFF], 2)
    assert not RF.embeddable([1, 2, 3, 64], 3) and RF.embeddable([31, 31, 31, 63], 3)


def test_malformed_blocks_are_rejected():
    rng = np.random.default_rng(9)
    docs, tfs = _case(rng, 50, 65536, 5, False)
    bid, ctp, cnt, pivot, body = RF.encode_term(docs, tfs, rng)[0]
    assert _decode((bid, ctp, cnt, pivot, body[:-3]))[0] < 0          # truncated container
    assert _decode((bid, ctp, cnt + 1, pivot, body))[0] < 0           # count does not match the container
    assert _decode((bid, (ctp & 0x3FFFFFFF), cnt, pivot, body))[0] < 0  # Delta containers are not written by the reference
    assert _decode((bid, ctp, cnt, pivot, body))[0] == cnt


@pytest.mark.gpu
def test_upload_ref_blocks_equals_csr_upload():
    from oracle import oracle as O
    rng = np.random.default_rng(11)
    n_docs = 200_000
    dl = O.lex_doclen(n_docs)
    dfs = [150_000, 20_000, 900, 40_000]
    lists = []
    for df in dfs:
        docs = np.sort(rng.choice(n_docs, size=df, replace=False)).astype(np.uint32)
        tfs = np.minimum(rng.geometric(0.45, size=df), 600).astype(np.uint16)
        lists.append((docs, tfs))
    base = bytes(rng.integers(0, 256, size=77, dtype=np.uint8))
    term_blocks = [RF.encode_term(d, t, rng, base_bytes=base) for d, t in lists]
    a, b = S.Shard(0), S.Shard(0)
    a.upload_ref_blocks(n_docs, dl, term_blocks)
    offs = np.zeros(len(lists) + 1, np.uint64)
    offs[1:] = np.cumsum(dfs)
    b.upload_lexical(n_docs, dl, offs, np.concatenate([d for d, _ in lists]), np.concatenate([t for _, t in lists]))
    osh = O.Shard(n_docs, dl, offs, np.concatenate([d for d, _ in lists]), np.concatenate([t for _, t in lists]))
    for qt, op, terms in ((S.QueryType.Union, O.OP_OR, [0, 1, 2]), (S.QueryType.Intersection, O.OP_AND, [0, 3]),
                          (S.QueryType.Union, O.OP_OR, [2]), (S.QueryType.Intersection, O.OP_AND, [0, 1, 3])):
        qa, qb = a.make_queries([terms], qt), b.make_queries([terms], qt)
        ra, rb_ = a.search_lexical_batch(qa, 10), b.search_lexical_batch(qb, 10)
        for x, y in zip(ra, rb_):
            assert np.array_equal(x, y)
        od, os_, otot = osh.search_exhaustive(terms, op, 10)
        assert int(ra[3][0]) == otot
        assert np.allclose(ra[1][0][:ra[2][0]], os_, rtol=1e-4)
    a.close()
    b.close()


# ------------------------------------------------------------------------------------------------ index.bin / vector.bin
def _corpus(rng, n_docs, dfs):
    from oracle import oracle as O
    dl = O.lex_doclen(n_docs)
    lists = []
    for df in dfs:
        docs = np.sort(rng.choice(n_docs, size=df, replace=False)).astype(np.uint32)
        tfs = np.minimum(rng.geometric(0.45, size=df), 600).astype(np.uint16)
        lists.append((docs, tfs))
    keys = sorted(int(k) & ~7 for k in rng.integers(1 << 40, 1 << 63, size=len(dfs), dtype=np.int64))
    assert len(set(keys)) == len(keys)
    perm = rng.permutation(len(dfs))  # key order != the order the lists were made in
    terms = [(keys[i], *lists[perm[i]]) for i in range(len(dfs))]
    return dl, terms


def _ngram_terms(rng, n_docs, head, dfs=(2_500, 70_000)):
    """real n-gram keys: bigram types for 22-byte heads, a trigram type too for 23-byte heads (index.rs:2806-2812)"""
    from oracle import oracle as O
    out = []
    types = [1, 3] if head == 22 else [2, 5]
    for df, ty in zip(dfs, types):
        docs = np.sort(rng.choice(n_docs, size=df, replace=False)).astype(np.uint32)
        counts = np.minimum(rng.geometric(0.6, size=df), 40).astype(np.uint16)          # occurrences of the n-gram itself
        nc = RF.ngram_components(ty)
        comp = (counts[:, None] + rng.integers(0, 300, size=(df, nc))).astype(np.uint32)  # each component occurs at least as often
        comp[rng.integers(0, df, 5), rng.integers(0, nc, 5)] = 20_000                     # 3-byte VINT
        key = (int(rng.integers(1 << 40, 1 << 62)) & ~7) | ty
        df_bytes = [O.lib().so_int_to_byte4(int(x)) for x in rng.integers(df, n_docs, nc)]
        out.append((key, docs, counts, comp, df_bytes))
    return out


@pytest.mark.parametrize("head,bits", [(20, 11), (22, 4), (23, 4)])
def test_index_bin_walk(head, bits):
    from oracle import oracle as O
    rng = np.random.default_rng(head)
    n_docs = 150_000  # 3 levels, the last incomplete
    dl, terms = _corpus(rng, n_docs, [40_000, 3_000, 17, 90_000, 1])
    ngram = _ngram_terms(rng, n_docs, head) if head != 20 else []
    data = RF.write_index_bin(n_docs, dl, terms, rng, segment_number_bits=bits, key_head_size=head, ngram_terms=ngram)
    ix = S.IndexBin(data, 1, head, bits)
    # every key in hash order; an n-gram key once per component term
    want = sorted([(t[0], 0, 1, t[1], t[2], 0) for t in terms] +
                  [(g[0], c, len(g[4]), g[1], g[3][:, c], O.lib().so_byte4_to_int(g[4][c])) for g in ngram for c in range(len(g[4]))],
                  key=lambda e: (e[0], e[1]))
    assert ix.indexed_doc_count == n_docs and ix.level_count == 3 and ix.term_count == len(want)
    assert ix.ngram_keys_skipped == 0
    assert [int(k) for k in ix.term_keys] == [w[0] for w in want]
    for t, (key, comp, ncomp, docs, tfs, df) in enumerate(want):
        assert ix.term_of_key(key) == t - comp
        assert (ix.term_components[t], ix.term_component[t], ix.term_component_df[t]) == (ncomp, comp, df)
        d, f = ix.postings(t)
        assert np.array_equal(d, docs) and np.array_equal(f, tfs)
    for g in ngram:  # what a query term that resolved to the n-gram contributes: its components with idf_ngram_i
        t0 = ix.term_of_key(g[0])
        got = ix.terms_of_key(g[0])
        assert [t for t, _ in got] == list(range(t0, t0 + len(g[4])))
        for (t, w), b in zip(got, g[4]):
            assert w == float(S.idf_f32(n_docs, O.lib().so_byte4_to_int(b)))
    assert ix.terms_of_key(terms[0][0]) == [(ix.term_of_key(terms[0][0]), None)]
    assert ix.term_of_key(terms[0][0] + 8) is None
    ix.close()
    # frequent terms only: the tail of the vocabulary is dropped and the term ids re-ranked
    ix = S.IndexBin(data, 1, head, bits, min_posting_count=1000)
    kept = [w for w in want if len(w[3]) >= 1000]
    assert ix.term_count == len(kept) == 3 + sum(len(g[4]) for g in ngram) and [int(k) for k in ix.term_keys] == [w[0] for w in kept]
    for t, (key, comp, ncomp, docs, tfs, df) in enumerate(kept):
        d, f = ix.postings(t)
        assert np.array_equal(d, docs) and np.array_equal(f, tfs)
        assert (ix.term_components[t], ix.term_component[t]) == (ncomp, comp)
    assert all(ix.term_of_key(t[0]) is None for t in terms if len(t[1]) < 1000)
    ix.close()


def test_ngram_block_decoder():
    """records of an n-gram key: component tfs before the positions count, never embedded (index_posting.rs:445, 666-722)"""
    rng = np.random.default_rng(7)
    docs = np.sort(rng.choice(65536, size=3000, replace=False))
    counts = rng.integers(1, 4, size=3000)
    for nc in (2, 3):
        comp = rng.integers(1, 1000, size=(3000, nc))
        comp[:50] = rng.integers(128, 30_000, size=(50, nc))  # 2- and 3-byte values
        for limit in (32768, 600):                            # 600: most postings behind 3-byte pointers
            blk = RF.encode_term(docs, counts, rng, positions_limit=limit, ngram_tfs=comp)[0]
            bid, ctp, cnt, pivot, body = blk
            assert limit == 32768 or pivot < cnt
            buf = np.frombuffer(body, np.uint8).copy()
            rb = N.RefBlock(bid, ctp, cnt - 1, pivot, buf.ctypes.data, len(buf))
            d = np.zeros(65536, np.uint16)
            t = np.zeros(65536, np.uint16)
            for c in range(nc):
                n = N.lib().ss_ref_decode_block_ngram(C.byref(rb), nc, c, N.ptr(d, N.u16p), N.ptr(t, N.u16p))
                assert n == 3000 and np.array_equal(d[:n], docs) and np.array_equal(t[:n], comp[:, c])
            assert N.lib().ss_ref_decode_block_ngram(C.byref(rb), nc, nc, N.ptr(d, N.u16p), N.ptr(t, N.u16p)) < 0
            assert N.lib().ss_ref_decode_block_ngram(C.byref(rb), 1, 0, N.ptr(d, N.u16p), N.ptr(t, N.u16p)) < 0
    # a SingleTerm block with embedded pointers is not an n-gram block
    blk = RF.encode_term([3, 9], [1, 2], rng)[0]
    buf = np.frombuffer(blk[4], np.uint8).copy()
    rb = N.RefBlock(blk[0], blk[1], blk[2] - 1, blk[3], buf.ctypes.data, len(buf))
    assert N.lib().ss_ref_decode_block_ngram(C.byref(rb), 2, 0, N.ptr(d, N.u16p), N.ptr(t, N.u16p)) < 0


def test_index_bin_rejects_garbage():
    rng = np.random.default_rng(3)
    dl, terms = _corpus(rng, 70_000, [500, 9])
    data = RF.write_index_bin(70_000, dl, terms, rng, segment_number_bits=3)
    for bad in (data[:len(data) - 5], b"\x05\x00\x01\x00" + data[4:], data[:3]):
        with pytest.raises(S.SeekStormHipError):
            S.IndexBin(bad, 1, 20, 3)
    empty = S.IndexBin(data[:4], 1, 20, 3)  # a freshly created index: header only (index.rs:2839-2853)
    assert empty.level_count == 0 and empty.term_count == 0
    with pytest.raises(S.SeekStormHipError):
        S.IndexBin(data, 1, 21, 3)


@pytest.mark.gpu
def test_upload_index_bin_and_vector_bin_answer_like_the_arrays():
    from oracle import oracle as O
    rng = np.random.default_rng(21)
    n_docs = 140_000
    dl, terms = _corpus(rng, n_docs, [60_000, 9_000, 700, 30_000])
    data = RF.write_index_bin(n_docs, dl, terms, rng)  # the reference's 2048 segments
    ix = S.IndexBin(data)
    a, b = S.Shard(0), S.Shard(0)
    a.upload_index_bin(ix)
    offs = np.zeros(len(terms) + 1, np.uint64)
    offs[1:] = np.cumsum([len(t[1]) for t in terms])
    alld, allt = np.concatenate([t[1] for t in terms]), np.concatenate([t[2] for t in terms])
    b.upload_lexical(n_docs, dl, offs, alld, allt)
    osh = O.Shard(n_docs, dl, offs, alld, allt)
    assert a.lexical_info() == b.lexical_info()
    for qt, op, q in ((S.QueryType.Union, O.OP_OR, [0, 1, 2]), (S.QueryType.Intersection, O.OP_AND, [0, 3]),
                      (S.QueryType.Union, O.OP_OR, [3, 1])):
        ra = a.search_lexical_batch(a.make_queries([q], qt), 10)
        rb_ = b.search_lexical_batch(b.make_queries([q], qt), 10)
        for x, y in zip(ra, rb_):
            assert np.array_equal(x, y)
        od, os_, otot = osh.search_exhaustive(q, op, 10)
        assert int(ra[3][0]) == otot and np.allclose(ra[1][0][:ra[2][0]], os_, rtol=1e-4)
    # vector.bin: 2 levels, clustered records, several records per doc in level 1
    dim = 64
    rows = O.vec_gen(O.VEC_SEED, 0, 700, dim)
    recs = [(int(i % 300), 0, int(i // 300), rows[i]) for i in range(700)]
    levels = [[recs[0:120], recs[120:300]], [recs[300:301], recs[301:650], recs[650:700]]]
    ids = np.array([(0 << 16) | r[0] for r in recs[:300]] + [(1 << 16) | r[0] for r in recs[300:]], np.uint32)
    a.upload_vector_bin(RF.write_vector_bin(levels, dim), dim)
    b.upload_vectors(rows, ids)
    qs = O.vec_gen(O.VECQ_SEED, 0, 5, dim)
    for x, y in zip(a.search_vector_batch(qs, 20), b.search_vector_batch(qs, 20)):
        assert np.array_equal(x, y)
    assert np.array_equal(a.read_rows(0, 700), rows)
    a.close()
    b.close()


@pytest.mark.gpu
def test_ngram_keys_of_an_index_bin_score_like_the_reference_arm():
    """the default index (NgramFF | NgramFFF, 23-byte key heads): a query term that resolved to an n-gram key is searched
    as the key's component terms with idf_ngram_i -- same matches, score = the n-gram arm of
    get_bm25f_multiterm_singlefield (add_result.rs:1454-1477)"""
    from oracle import oracle as O
    rng = np.random.default_rng(33)
    n_docs, head = 140_000, 23
    dl, terms = _corpus(rng, n_docs, [50_000, 8_000, 20_000])
    ngram = _ngram_terms(rng, n_docs, head, dfs=(6_000, 30_000))
    data = RF.write_index_bin(n_docs, dl, terms, rng, key_head_size=head, ngram_terms=ngram)
    ix = S.IndexBin(data, 1, head)
    sh = S.Shard(0)
    sh.upload_index_bin(ix)
    # the oracle sees one posting list per device term: single terms as they are, n-gram keys as their components
    lists = []
    for t in range(ix.term_count):
        d, f = ix.postings(t)
        lists.append((d, f))
    offs = np.zeros(len(lists) + 1, np.uint64)
    offs[1:] = np.cumsum([len(l[0]) for l in lists])
    osh = O.Shard(n_docs, dl, offs, np.concatenate([l[0] for l in lists]), np.concatenate([l[1] for l in lists]))
    single = [ix.term_of_key(t[0]) for t in terms]
    grams = [ix.terms_of_key(g[0]) for g in ngram]
    assert [len(g) for g in grams] == [2, 3]
    for qt, op, keys in ((S.QueryType.Union, O.OP_OR, [grams[0], [(single[0], None)]]),
                         (S.QueryType.Intersection, O.OP_AND, [grams[1], [(single[0], None)]]),
                         (S.QueryType.Union, O.OP_OR, [grams[1]]),
                         (S.QueryType.Intersection, O.OP_AND, [grams[0], grams[1], [(single[2], None)]])):
        tl = [t for g in keys for t, _ in g]
        idf_of = {t: w for g in keys for t, w in g if w is not None}
        idf = [idf_of.get(t, float(S.idf_f32(n_docs, len(lists[t][0])))) for t in tl]
        doc, score, cnt, tot = sh.search_lexical_batch(sh.make_queries([tl], qt, idf_of=idf_of), 10)
        od, os_, otot = osh.search_exhaustive(tl, op, 10, idf=idf)
        n = int(cnt[0])
        assert int(tot[0]) == otot and n == len(od)
        assert np.allclose(score[0][:n], os_, rtol=1e-4)
        assert {int(x) for x, y in zip(doc[0][:n], score[0]) if y > os_[-1] * (1 + 1e-4)} == \
               {int(x) for x, y in zip(od, os_) if y > os_[-1] * (1 + 1e-4)}
    # the arm itself, by hand, for the best hit of the lone trigram query
    tl = [t for t, _ in grams[1]]
    idf_of = dict(grams[1])
    doc, score, cnt, tot = sh.search_lexical_batch(sh.make_queries([tl], S.QueryType.Union, idf_of=idf_of), 1)
    d0 = int(doc[0][0])
    comp = np.zeros(256, np.float32)
    O.lib().so_bm25_component_cache(osh.avgdl() if callable(osh.avgdl) else osh.avgdl, comp.ctypes.data_as(C.POINTER(C.c_float)))
    g = ngram[1]
    r = int(np.searchsorted(g[1], d0))
    want = np.float32(0)
    for c in range(3):
        tf = np.float32(g[3][r, c])
        want += np.float32(idf_of[tl[c]]) * (tf * np.float32(2.2) / (tf + comp[dl[d0]]))
    assert abs(float(want) - float(score[0][0])) <= 1e-4 * float(want)
    sh.close()
    ix.close()


@pytest.mark.gpu
def test_vector_bin_i8_records_and_shard_seam():
    """Precision::I8 vector.bin (header + dim x i8, VectorHeader.scale per record) == the array upload; the shard seam
    quantises the query like the reference (quantize_f32_to_i8) and reports the raw integer dot"""
    from oracle import oracle as O
    dim = 96
    rows = O.quantize_i8(O.vec_gen(O.VEC_SEED, 0, 500, dim))
    rng = np.random.default_rng(2)
    sc = rng.uniform(0.01, 0.02, 500).astype(np.float32)
    recs = [(int(i % 250), 0, int(i // 250), rows[i], float(sc[i])) for i in range(500)]
    levels = [[recs[0:100], recs[100:250]], [recs[250:500]]]
    ids = np.array([r[0] for r in recs[:250]] + [(1 << 16) | r[0] for r in recs[250:]], np.uint32)
    data = RF.write_vector_bin(levels, dim, i8=True)
    qs = O.quantize_i8(O.vec_gen(O.VECQ_SEED, 0, 4, dim))
    a, b = S.Shard(0), S.Shard(0)
    for use_scale in (False, True):
        a.upload_vector_bin(data, dim, i8=True, use_record_scale=use_scale)
        b.upload_vectors_i8(rows, row_scale=sc if use_scale else None, row_doc_ids=ids)
        assert np.array_equal(a.read_rows_i8(0, 500), rows)
        qscale = np.full(4, 0.5, np.float32) if use_scale else None
        for x, y in zip(a.search_vector_batch_i8(qs, 20, query_scale=qscale), b.search_vector_batch_i8(qs, 20, query_scale=qscale)):
            assert np.array_equal(x, y)
    a.upload_vector_bin(data, dim, i8=True)
    qf = O.vec_gen(O.VECQ_SEED, 0, 1, dim)[0]
    ro = a.search_vector_shard(qf, 10)
    od, os_, _, _ = O.vec_search_i8(rows, O.quantize_i8(qf), 10, row_doc_ids=ids)
    assert [r.score for r in ro.results] == [float(x) for x in os_] and ro.results[0].doc_id == int(od[0])
    a.close()
    b.close()


# ------------------------------------------------------------------------------------------------ several indexed fields
def _decode_fields(block, n_fields, longest):
    bid, ctp, cnt, pivot, body = block
    buf = np.frombuffer(body, np.uint8).copy()
    rb = N.RefBlock(bid, ctp, cnt - 1, pivot, buf.ctypes.data, len(buf))
    d = np.zeros(65536, np.uint16); first = np.zeros(65537, np.uint32)
    f = np.zeros(65536 * n_fields, np.uint8); t = np.zeros(65536 * n_fields, np.uint16)
    n = N.lib().ss_ref_decode_block_fields(C.byref(rb), n_fields, longest, N.ptr(d, N.u16p), N.ptr(first, N.u32p), N.ptr(f, N.u8p),
                                           N.ptr(t, N.u16p))
    return n, d, first, f, t


def _fields_postings(rng, n, n_fields, tf_hi, longest):
    docs = np.sort(rng.choice(65536, size=n, replace=False))
    d, f, t = [], [], []
    for doc in docs:
        k = int(rng.integers(1, n_fields + 1)) if rng.random() < 0.5 else 1
        fs = np.sort(rng.choice(n_fields, size=k, replace=False)) if rng.random() < 0.7 else np.array([longest])
        for x in fs:
            d.append(int(doc)); f.append(int(x)); t.append(int(rng.integers(1, tf_hi + 1)))
    return np.array(d), np.array(f), np.array(t)


@pytest.mark.parametrize("n_fields,longest,n,tf_hi,limit", [(2, 0, 300, 3, 32768), (3, 1, 2000, 6, 32768), (4, 3, 500, 300, 32768),
                                                           (3, 0, 900, 5, 96), (8, 5, 700, 4, 32768), (2, 1, 5000, 2, 32768)])
def test_decode_fields_roundtrip(n_fields, longest, n, tf_hi, limit):
    """decode_positions_multiterm_multifield / read_multifield_vec (add_result.rs:1485-2034, 2200-2293): embedded 2- and
    3-byte forms (longest-field, one field, two and three fields) and field vectors in front of VINT records"""
    rng = np.random.default_rng(n_fields * 100 + n)
    d, f, t = _fields_postings(rng, n, n_fields, tf_hi, longest)
    blocks = RF.encode_term_fields(d, f, t, n_fields, longest, rng, positions_limit=limit, max_gap=25)
    assert len(blocks) == 1
    cnt, dd, first, ff, tt = _decode_fields(blocks[0], n_fields, longest)
    assert cnt == len(np.unique(d)) and int(first[cnt]) == len(d)
    assert np.array_equal(np.repeat(dd[:cnt], np.diff(first[:cnt + 1])), d)
    assert np.array_equal(ff[:len(d)], f) and np.array_equal(tt[:len(d)], t)
    if limit < 32768:
        assert 0 < blocks[0][3] < cnt  # 3-byte pointers in use


def _decode_fields_positions(block, n_fields, longest, cap=1 << 22):
    bid, ctp, cnt, pivot, body = block
    buf = np.frombuffer(body, np.uint8).copy()
    rb = N.RefBlock(bid, ctp, cnt - 1, pivot, buf.ctypes.data, len(buf))
    d = np.zeros(65536, np.uint16); first = np.zeros(65537, np.uint32)
    f = np.zeros(65536 * n_fields, np.uint8); t = np.zeros(65536 * n_fields, np.uint16)
    pos = np.zeros(cap, np.uint16)
    npos = np.zeros(1, np.uint64)
    n = N.lib().ss_ref_decode_block_fields_positions(C.byref(rb), n_fields, longest, N.ptr(d, N.u16p), N.ptr(first, N.u32p),
                                                     N.ptr(f, N.u8p), N.ptr(t, N.u16p), N.ptr(pos, N.u16p), cap, N.ptr(npos, N.u64p))
    return n, d, first, f, t, pos[:int(npos[0])]


@pytest.mark.parametrize("n_fields,longest,n,tf_hi,limit,gap", [(2, 0, 300, 3, 32768, 25), (3, 1, 2000, 6, 32768, 6), (4, 3, 500, 300, 32768, 40),
                                                               (3, 0, 900, 5, 96, 3), (8, 5, 700, 4, 32768, 9), (2, 1, 5000, 2, 32768, 600),
                                                               (2, 0, 1500, 4, 32768, 2), (5, 2, 1200, 3, 200, 30), (3, 1, 800, 3, 32768, 21000)])
def test_decode_fields_positions_roundtrip(n_fields, longest, n, tf_hi, limit, gap):
    """the POSITIONS of a multi-field index (decode_positions_multiterm_multifield add_result.rs:1485-2034 +
    get_next_position_multifield): VINT positions behind a record's field vector, and the bit-packed positions of every embedded
    form (2 bytes: 13 bits longest-field, 12 - id bits per named field; 3 bytes: 20 / 19 - id bits), restarting in every field"""
    rng = np.random.default_rng(n_fields * 1000 + n + gap)
    d, f, t = _fields_postings(rng, n, n_fields, tf_hi, longest)
    positions = [RF.random_positions(rng, int(x), gap) for x in t]
    blocks = RF.encode_term_fields(d, f, t, n_fields, longest, rng, positions_limit=limit, positions=positions)
    assert len(blocks) == 1
    cnt, dd, first, ff, tt, pos = _decode_fields_positions(blocks[0], n_fields, longest)
    assert cnt == len(np.unique(d)) and int(first[cnt]) == len(d)
    assert np.array_equal(ff[:len(d)], f) and np.array_equal(tt[:len(d)], t)
    assert np.array_equal(pos, np.concatenate([np.asarray(p, np.uint16) for p in positions]))
    # both pointer forms and both record kinds occurred
    body = blocks[0][4]
    assert len(body) > 0


def test_embedded_field_pointer_tags():
    # 2-byte tags (bits 15..12): 110x / 111x longest field 1 / 2 positions; 1000 / 1001 / 1010 one field 1..3; 1011 two fields
    # 3-byte tags (bits 23..19): 1100x..1111x longest 1..4; 10000..10011 one field 1..4; 10100 (1,1) 10101 (1,2) 10110 (2,1) 10111 (1,1,1)
    E = RF.embed_fields
    assert E([(1, [5])], True, 1, 2)[1] >> 4 in (0xC, 0xD) and E([(1, [5, 6])], True, 1, 2)[1] >> 4 in (0xE, 0xF)
    assert E([(2, [5])], False, 2, 2)[1] >> 4 == 0x8 and E([(2, [1, 2, 3])], False, 2, 2)[1] >> 4 == 0xA
    assert E([(0, [1]), (2, [3])], False, 2, 2)[1] >> 4 == 0xB
    assert E([(1, [9, 9, 9, 9])], True, 1, 3)[2] >> 3 in (0x1E, 0x1F) and E([(1, [7, 7, 7, 7])], False, 2, 3)[2] >> 3 == 0x13
    assert E([(0, [1]), (1, [2, 3])], False, 2, 3)[2] >> 3 == 0x15 and E([(0, [1, 2]), (1, [3])], False, 2, 3)[2] >> 3 == 0x16
    assert E([(0, [1]), (1, [2]), (2, [3])], False, 2, 3)[2] >> 3 == 0x17


@pytest.mark.gpu
def test_upload_index_bin_with_several_fields():
    from oracle import oracle as O
    rng = np.random.default_rng(41)
    n_docs, n_fields, longest = 100_000, 3, 1
    dl = np.stack([O.lex_doclen(n_docs, seed=O.LEX_SEED + 5 * f) for f in range(n_fields)])
    keys = sorted(int(k) & ~7 for k in rng.integers(1 << 40, 1 << 63, size=4, dtype=np.int64))
    terms, offs, D, F, T = [], [0], [], [], []
    for key, df in zip(keys, (20_000, 700, 6_000, 40)):
        docs = np.sort(rng.choice(n_docs, size=df, replace=False))
        d, f, t = [], [], []
        for doc in docs:
            fs = np.sort(rng.choice(n_fields, size=int(rng.integers(1, n_fields + 1)), replace=False)) if rng.random() < 0.6 else [longest]
            for x in fs:
                d.append(int(doc)); f.append(int(x)); t.append(int(min(rng.geometric(0.5), 30)))
        terms.append((key, np.array(d), np.array(f), np.array(t)))
        D += d; F += f; T += t
        offs.append(len(D))
    data = RF.write_index_bin(n_docs, dl, terms, rng, n_fields=n_fields, longest_field_id=longest)
    ix = S.IndexBin(data, n_fields)
    assert ix.indexed_doc_count == n_docs and ix.term_count == 4
    boost = [1.5, 1.0, 0.5]
    a, b = S.Shard(0), S.Shard(0)
    a.upload_index_bin(ix, boost)
    b.upload_lexical_fields(n_docs, dl, boost, np.array(offs, np.uint64), np.array(D, np.uint32), np.array(F, np.uint8), np.array(T, np.uint16))
    assert a.lexical_info() == b.lexical_info()
    for qt, oop in ((S.QueryType.Union, O.OP_OR), (S.QueryType.Intersection, O.OP_AND)):
        for q in ([0, 2], [1], [0, 1, 2], [3, 0]):
            ra = a.search_lexical_batch(a.make_queries([q], qt), 10)
            rb_ = b.search_lexical_batch(b.make_queries([q], qt), 10)
            for x, y in zip(ra, rb_):
                assert np.array_equal(x, y)
            od, os_, otot, _ = O.search_fields_exhaustive(n_docs, dl, boost, offs, D, F, T, q, oop, 10)
            assert int(ra[3][0]) == otot and np.allclose(ra[1][0][:ra[2][0]], os_, rtol=1e-4)
    a.close()
    b.close()


def test_decoders_survive_corrupted_bytes():
    """bit flips, truncations and random headers: every reader returns a result or an error code, never walks outside the
    byte array (run under the CPU suite: a crash here would take the process down)"""
    rng = np.random.default_rng(77)
    d1, t1 = _case(rng, 400, 65536, 40, False)
    one = RF.encode_term(d1, t1, rng, base_bytes=bytes(50))[0]
    d, f, t = _fields_postings(rng, 400, 3, 40, 1)
    many = RF.encode_term_fields(d, f, t, 3, 1, rng, max_gap=25)[0]
    for blk, fields in ((one, False), (many, True)):
        bid, ctp, cnt, pivot, body = blk
        for trial in range(1500):
            b = bytearray(body)
            kind = trial % 5
            if kind == 0:
                for _ in range(int(rng.integers(1, 6))):
                    b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
                blk2 = (bid, ctp, cnt, pivot, bytes(b))
            elif kind == 1:
                blk2 = (bid, ctp, cnt, pivot, bytes(b[:int(rng.integers(0, len(b)))]))
            elif kind == 2:
                blk2 = (bid, int(rng.integers(0, 1 << 32)), cnt, pivot, bytes(b))
            elif kind == 3:
                blk2 = (bid, ctp, int(rng.integers(1, 65537)), int(rng.integers(0, 65536)), bytes(b))
            else:
                blk2 = (bid, ctp, cnt, pivot, bytes(rng.integers(0, 256, size=len(b), dtype=np.uint8)))
            if not blk2[4]:
                continue
            n = (_decode_fields(blk2, 3, 1) if fields else _decode(blk2))[0]
            assert n <= 65536
            # ... and the position decoders: never more positions than the bytes (+ 4 embedded per posting) can hold, however the
            # corrupted pointers overlap
            if fields:
                r = _decode_fields_positions(blk2, 3, 1)
                assert r[0] <= 65536 and len(r[-1]) <= len(blk2[4]) + 4 * 65536
            else:
                buf = np.frombuffer(blk2[4], np.uint8).copy()
                rb = N.RefBlock(blk2[0], blk2[1], blk2[2] - 1, blk2[3], buf.ctypes.data, len(buf))
                dd, tt, pp, npos = np.zeros(65536, np.uint16), np.zeros(65536, np.uint16), np.zeros(1 << 20, np.uint16), C.c_uint64()
                n2 = N.lib().ss_ref_decode_block_positions(C.byref(rb), N.ptr(dd, N.u16p), N.ptr(tt, N.u16p), N.ptr(pp, N.u16p), len(pp), C.byref(npos))
                assert n2 <= 65536 and (n2 < 0 or npos.value <= len(buf) + 4 * 65536)
    # index.bin / vector headers made of noise
    for trial in range(200):
        junk = b"\\x06\\x00\\x01\\x00" + bytes(rng.integers(0, 256, size=int(rng.integers(0, 300_000)), dtype=np.uint8))
        try:
            S.IndexBin(junk, int(rng.integers(1, 4)), 20, int(rng.integers(0, 5))).close()
        except S.SeekStormHipError:
            pass


def test_golden_key_bodies():
    """committed byte arrays (tests/golden/make_ref_format_golden.py) decode to their committed postings"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_format.npz"))
    for name in ("array", "bitmap", "rle", "pivot"):
        bid, ctp, cnt, pivot = (int(x) for x in g[f"s_{name}_head"])
        n, d, t = _decode((bid, ctp, cnt, pivot, g[f"s_{name}_body"].tobytes()))
        assert n == cnt and np.array_equal(d, g[f"s_{name}_docs"]) and np.array_equal(t, g[f"s_{name}_tfs"])
        assert ctp >> 30 == {"array": RF.CT_ARRAY, "bitmap": RF.CT_BITMAP, "rle": RF.CT_RLE, "pivot": RF.CT_ARRAY}[name]
    bid, ctp, cnt, pivot, nf, longest = (int(x) for x in g["m_head"])
    n, dd, first, ff, tt = _decode_fields((bid, ctp, cnt, pivot, g["m_body"].tobytes()), nf, longest)
    m = len(g["m_docs"])
    assert n == cnt and int(first[cnt]) == m
    assert np.array_equal(np.repeat(dd[:cnt], np.diff(first[:cnt + 1])), g["m_docs"])
    assert np.array_equal(ff[:m], g["m_fields"]) and np.array_equal(tt[:m], g["m_tfs"])
    # n-gram key: every component's tf
    bid, ctp, cnt, pivot, nc = (int(x) for x in g["g_head"])
    buf = g["g_body"].copy()
    rb = N.RefBlock(bid, ctp, cnt - 1, pivot, buf.ctypes.data, len(buf))
    d = np.zeros(65536, np.uint16)
    t = np.zeros(65536, np.uint16)
    assert 0 < pivot < cnt
    for c in range(nc):
        assert N.lib().ss_ref_decode_block_ngram(C.byref(rb), nc, c, N.ptr(d, N.u16p), N.ptr(t, N.u16p)) == cnt
        assert np.array_equal(d[:cnt], g["g_docs"]) and np.array_equal(t[:cnt], g["g_tfs"][:, c])


def test_hand_assembled_key_bodies():
    """Key bodies assembled by hand from the reference's WRITERS (tests/golden/hand_assembled_blocks.py: every byte cites
    index_posting.rs / compress_postinglist.rs), not by oracle/ref_format.py: Array / Bitmap / Rle containers, 2- and 3-byte
    pointers with the pivot inside the list, every single-field embedded form, 1- and 2-byte position counts."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("hand_assembled_blocks", os.path.join(os.path.dirname(__file__), "golden", "hand_assembled_blocks.py"))
    H = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(H)
    assert len(H.BLOCKS) >= 4
    for b in H.BLOCKS:
        blk = (b["block_id"], b["compression_type_pointer"], b["posting_count"], b["pointer_pivot_p_docid"], b["body"])
        cnt, d, t = _decode(blk)
        assert cnt == len(b["docs"]), (b["name"], cnt)
        assert d.tolist() == b["docs"] and t.tolist() == b["tfs"], (b["name"], d.tolist(), t.tolist())
        cnt2, d2, t2, pos = _decode_positions(blk)  # the positions a phrase query walks, as absolute values
        assert cnt2 == cnt and pos.tolist() == b["positions"], (b["name"], pos.tolist())
        # and the restated writer, given the same postings, chooses bytes the decoder reads the same way (two routes, one answer)
    # the fixture's own arithmetic: pointer ranges
    assert H.H1_R == 144 and H.H2_R == 9 and H.H4_R == 15
    # several indexed fields: field vectors, per-field positions, every embedded form with named fields (H5: 2-byte pointers,
    # H6: 3-byte pointers) -- the bytes follow index_posting.rs:592-660 / write_field_vec 897-925, the decoder follows
    # add_result.rs:1485-2034
    assert len(H.FIELD_BLOCKS) >= 2
    for b in H.FIELD_BLOCKS:
        blk = (b["block_id"], b["compression_type_pointer"], b["posting_count"], b["pointer_pivot_p_docid"], b["body"])
        cnt, dd, first, ff, tt, pos = _decode_fields_positions(blk, b["n_fields"], b["longest_field_id"])
        assert cnt == len(b["docs"]) and dd[:cnt].tolist() == b["docs"], (b["name"], cnt)
        flat = [(f, p) for e in b["entries"] for f, p in e]
        assert first[:cnt + 1].tolist() == np.concatenate([[0], np.cumsum([len(e) for e in b["entries"]])]).tolist(), b["name"]
        assert ff[:len(flat)].tolist() == [f for f, _ in flat] and tt[:len(flat)].tolist() == [len(p) for _, p in flat], b["name"]
        assert pos.tolist() == [x for _, p in flat for x in p], (b["name"], pos.tolist())
        # the restated writer produces the same bytes from the same postings (two routes, one answer)
        body, ctp, n, pivot = RF.encode_key_body_fields(b["docs"], b["entries"], b["n_fields"], b["longest_field_id"],
                                                        positions_limit=32768 if b["pointer_pivot_p_docid"] else 0)
        if b["pointer_pivot_p_docid"]:
            assert body == b["body"] and ctp == b["compression_type_pointer"] and pivot == b["pointer_pivot_p_docid"], b["name"]


def _decode_ngram_positions(block, nc):
    bid, ctp, cnt, pivot, body = block
    buf = np.frombuffer(body, np.uint8).copy()
    rb = N.RefBlock(bid, ctp, cnt - 1, pivot, buf.ctypes.data, len(buf))
    d, t, c = np.zeros(65536, np.uint16), np.zeros(65536, np.uint16), np.zeros(65536, np.uint16)
    npos = C.c_uint64()
    N.lib().ss_ref_decode_block_ngram_positions(C.byref(rb), nc, N.ptr(d, N.u16p), N.ptr(t, N.u16p), N.ptr(c, N.u16p), None, 0, C.byref(npos))
    pos = np.zeros(max(npos.value, 1), np.uint16)
    n = N.lib().ss_ref_decode_block_ngram_positions(C.byref(rb), nc, N.ptr(d, N.u16p), N.ptr(t, N.u16p), N.ptr(c, N.u16p), N.ptr(pos, N.u16p),
                                                    len(pos), C.byref(npos))
    return n, d[:max(n, 0)].copy(), t[:max(n, 0)].copy(), c[:max(n, 0)].copy(), pos[:npos.value].copy(), rb, buf


def test_hand_assembled_ngram_key_bodies():
    """H8 / H9 (tests/golden/hand_assembled_blocks.py): records of n-gram keys assembled from index_posting.rs:666-741 -- the
    component tfs, then the key's OWN positions_count and positions, which a phrase entry that resolved to the key walks"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("hand_assembled_blocks", os.path.join(os.path.dirname(__file__), "golden", "hand_assembled_blocks.py"))
    H = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(H)
    assert len(H.NGRAM_BLOCKS) == 2 and H.H8_R == 18 and H.H9_R == 13
    d16, t16 = np.zeros(65536, np.uint16), np.zeros(65536, np.uint16)
    for b in H.NGRAM_BLOCKS:
        blk = (b["block_id"], b["compression_type_pointer"], b["posting_count"], b["pointer_pivot_p_docid"], b["body"])
        nc = b["n_components"]
        n, d, t, c, pos, rb, buf = _decode_ngram_positions(blk, nc)
        assert n == len(b["docs"]) and d.tolist() == b["docs"], b["name"]
        assert t.tolist() == [x[0] for x in b["component_tfs"]] and c.tolist() == b["counts"] and pos.tolist() == b["positions"], (b["name"], pos.tolist())
        for comp in range(nc):
            assert N.lib().ss_ref_decode_block_ngram(C.byref(rb), nc, comp, N.ptr(d16, N.u16p), N.ptr(t16, N.u16p)) == n
            assert t16[:n].tolist() == [x[comp] for x in b["component_tfs"]], (b["name"], comp)
        # the restated writer, given the same postings, writes the same bytes behind a prefix of the same length (two routes, one answer)
        per_doc, at = [], 0
        for k_ in b["counts"]:
            per_doc.append(b["positions"][at:at + k_]); at += k_
        prefix = len(b["body"]) - len(RF.encode_key_body(b["docs"], per_doc, 0, 32768, b["component_tfs"])[0])
        if b["pointer_pivot_p_docid"] == b["posting_count"]:  # (H8 moves the pivot by hand: legal header, not the writer's choice)
            body, ctp, cnt, pivot = RF.encode_key_body(b["docs"], per_doc, prefix, 32768, b["component_tfs"])
            assert b["body"][prefix:] == body and ctp == b["compression_type_pointer"] and pivot == b["pointer_pivot_p_docid"], b["name"]


@pytest.mark.parametrize("limit", [32768, 400])
def test_ngram_positions_roundtrip(limit):
    """the key's own positions behind the component tfs: restated writer -> decoder (2- and 3-byte pointers)"""
    rng = np.random.default_rng(21)
    n = 2500
    docs = np.sort(rng.choice(65536, size=n, replace=False))
    counts = rng.integers(1, 6, size=n)
    per_doc = [RF.random_positions(rng, int(k_), 300) for k_ in counts]
    for nc in (2, 3):
        comp = rng.integers(1, 400, size=(n, nc))
        blk = RF.encode_term(docs, counts, rng, positions_limit=limit, ngram_tfs=comp, positions=per_doc)[0]
        assert limit == 32768 or blk[3] < blk[2]
        cnt, d, t, c, pos, _, _ = _decode_ngram_positions(blk, nc)
        assert cnt == n and np.array_equal(d, docs) and np.array_equal(t, comp[:, 0]) and np.array_equal(c, counts)
        assert pos.tolist() == [x for p_ in per_doc for x in p_]


@pytest.mark.parametrize("n,tf_hi,limit", [(40, 3, 32768), (3000, 6, 32768), (4000, 30, 32768), (500, 5, 600), (700, 3, 32769)])
def test_positions_roundtrip(n, tf_hi, limit):
    """ss_ref_decode_block_positions against the restated writer: embedded 2- and 3-byte forms, VINT records, the pivot inside
    the list (real and lowered limits), gaps that need 1- and 2-byte VINTs"""
    rng = np.random.default_rng(n + tf_hi)
    docs, tfs = _case(rng, n, 65536, tf_hi)
    # (limit 32769 marks the case with gaps of 16 384 and more: the three-byte position form, which keeps bit 13 twice)
    positions = [RF.random_positions(rng, int(tf), max_gap=int(rng.choice([3, 40, 300] if limit != 32769 else [300, 20000, 30000]))) for tf in tfs]
    blk = RF.encode_term(docs, tfs, rng, positions_limit=min(limit, 32768), positions=positions)[0]
    cnt, d, t, pos = _decode_positions(blk)
    assert cnt == len(docs) and np.array_equal(d, docs) and np.array_equal(t, tfs)
    assert pos.tolist() == [p for pl in positions for p in pl]


def _ngram_fields_corpus(rng, n, n_fields, longest, nc, span=65536):
    """an n-gram key of a multi-field index: per doc the n-gram's own (field, positions count) entries and, per component term,
    that term's field vector in the doc"""
    docs = np.sort(rng.choice(span, size=n, replace=False))
    d, f, t, vecs = [], [], [], {}
    for doc in docs:
        fs = np.sort(rng.choice(n_fields, size=int(rng.integers(1, n_fields + 1)), replace=False)) if rng.random() < 0.6 else np.array([longest])
        for x in fs:
            d.append(int(doc)); f.append(int(x)); t.append(int(min(rng.geometric(0.6), 20)))
        cv = []
        for c in range(nc):
            if rng.random() < 0.4:
                cf = [longest]
            else:
                cf = sorted(set(int(x) for x in fs) | set(int(x) for x in rng.choice(n_fields, size=int(rng.integers(1, n_fields + 1)), replace=False)))
            cv.append([(x, int(rng.choice([1, 3, 70, 9000]))) for x in cf])
        vecs[int(doc)] = cv
    return np.array(d), np.array(f), np.array(t), vecs


@pytest.mark.parametrize("n_fields,longest,nc,n,limit", [(2, 0, 2, 400, 32768), (3, 1, 3, 1500, 32768), (4, 3, 2, 900, 120), (8, 5, 3, 300, 32768)])
def test_decode_ngram_blocks_of_a_multi_field_index(n_fields, longest, nc, n, limit):
    """ss_ref_decode_block_fields_ngram: the records of an n-gram key in a multi-field index start with the field vector of
    each component term (index_posting.rs:664-722, add_result.rs:1524-1600); every component comes back as its own
    (field, tf) entries, 2- and 3-byte pointers, 1- to 3-byte vector entries"""
    rng = np.random.default_rng(n_fields * 31 + n)
    d, f, t, vecs = _ngram_fields_corpus(rng, n, n_fields, longest, nc)
    blk = RF.encode_term_fields(d, f, t, n_fields, longest, rng, positions_limit=limit, max_gap=25, ngram_vecs=vecs)[0]
    bid, ctp, cnt, pivot, body = blk
    buf = np.frombuffer(body, np.uint8).copy()
    rb = N.RefBlock(bid, ctp, cnt - 1, pivot, buf.ctypes.data, len(buf))
    docs = np.unique(d)
    for c in range(nc):
        dd = np.zeros(65536, np.uint16); first = np.zeros(65537, np.uint32)
        ff = np.zeros(65536 * n_fields, np.uint8); tt = np.zeros(65536 * n_fields, np.uint16)
        got = N.lib().ss_ref_decode_block_fields_ngram(C.byref(rb), n_fields, longest, nc, c, N.ptr(dd, N.u16p), N.ptr(first, N.u32p),
                                                       N.ptr(ff, N.u8p), N.ptr(tt, N.u16p))
        assert got == len(docs) and np.array_equal(dd[:got], docs)
        want = [e for doc in docs for e in vecs[int(doc)][c]]
        assert int(first[got]) == len(want)
        assert ff[:len(want)].tolist() == [e[0] for e in want] and tt[:len(want)].tolist() == [min(e[1], 65535) for e in want]
    if limit < 32768:
        assert 0 < pivot < cnt
    # the SingleTerm reader misreads such a block or refuses it; the n-gram reader refuses a component that is not there
    assert N.lib().ss_ref_decode_block_fields_ngram(C.byref(rb), n_fields, longest, nc, nc, None, None, None, None) == -1


@pytest.mark.gpu
def test_ngram_keys_of_a_multi_field_index_score_like_their_component_lists():
    """index.bin with several indexed fields AND n-gram keys (the default index over several fields): every n-gram key becomes
    one posting list per component term (its field vectors), scored with idf_ngram_i like the single-field arm -- the image
    built from the file answers like the array upload of the same (term / component, field) lists and like the oracle"""
    from oracle import oracle as O
    rng = np.random.default_rng(77)
    n_docs, n_fields, longest, head = 90_000, 3, 1, 23
    dl = np.stack([O.lex_doclen(n_docs, seed=O.LEX_SEED + 5 * f) for f in range(n_fields)])
    keys = sorted(int(k) & ~7 for k in rng.integers(1 << 40, 1 << 62, size=2, dtype=np.int64))
    terms = []
    for key, df in zip(keys, (15_000, 4_000)):
        docs = np.sort(rng.choice(n_docs, size=df, replace=False))
        d, f, t = [], [], []
        for doc in docs:
            fs = np.sort(rng.choice(n_fields, size=int(rng.integers(1, n_fields + 1)), replace=False)) if rng.random() < 0.6 else [longest]
            for x in fs:
                d.append(int(doc)); f.append(int(x)); t.append(int(min(rng.geometric(0.5), 30)))
        terms.append((key, np.array(d), np.array(f), np.array(t)))
    ngram = []
    for ty, df in ((2, 5_000), (5, 1_200)):  # a bigram and a trigram key
        nc = RF.ngram_components(ty)
        d, f, t, vecs = _ngram_fields_corpus(rng, df, n_fields, longest, nc, span=n_docs)
        key = (int(rng.integers(1 << 40, 1 << 62)) & ~7) | ty
        df_bytes = [O.lib().so_int_to_byte4(int(x)) for x in rng.integers(df, n_docs, nc)]
        ngram.append((key, d, f, t, vecs, df_bytes))
    data = RF.write_index_bin(n_docs, dl, terms, rng, key_head_size=head, n_fields=n_fields, longest_field_id=longest, ngram_terms=ngram)
    ix = S.IndexBin(data, n_fields, head)
    assert ix.term_count == 2 + 2 + 3
    boost = [1.5, 1.0, 0.5]
    a, b = S.Shard(0), S.Shard(0)
    a.upload_index_bin(ix, boost)
    # the same lists as arrays: device term ids follow the key order; an n-gram key contributes its components in order
    entries = sorted([(t_[0], "term", t_) for t_ in terms] + [(g[0], "gram", g) for g in ngram], key=lambda e: e[0])
    offs, D, F, T = [0], [], [], []
    for key, kind, e in entries:
        if kind == "term":
            D += e[1].tolist(); F += e[2].tolist(); T += e[3].tolist(); offs.append(len(D))
        else:
            docs = np.unique(e[1])
            for c in range(RF.ngram_components(key)):
                for doc in docs:
                    for fld, tf in e[4][int(doc)][c]:
                        D.append(int(doc)); F.append(fld); T.append(min(tf, 65535))
                offs.append(len(D))
    b.upload_lexical_fields(n_docs, dl, boost, np.array(offs, np.uint64), np.array(D, np.uint32), np.array(F, np.uint8), np.array(T, np.uint16))
    assert a.lexical_info() == b.lexical_info()
    grams = [ix.terms_of_key(g[0]) for g in ngram]
    assert sorted(len(g) for g in grams) == [2, 3]
    single = [ix.term_of_key(t_[0]) for t_ in terms]
    for qt in (S.QueryType.Union, S.QueryType.Intersection):
        for keyset in ([grams[0]], [grams[1], [(single[0], None)]], [grams[0], grams[1], [(single[1], None)]]):
            tl = [t_ for g in keyset for t_, _ in g]
            idf_of = {t_: w for g in keyset for t_, w in g if w is not None}
            ra = a.search_lexical_batch(a.make_queries([tl], qt, idf_of=idf_of), 10)
            rb_ = b.search_lexical_batch(b.make_queries([tl], qt, idf_of=idf_of), 10)
            for x, y in zip(ra, rb_):
                assert np.array_equal(x, y)
            assert int(ra[3][0]) > 0
    a.close()
    b.close()


def _hand_index_bin():
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("hand_assembled_index_bin", os.path.join(os.path.dirname(__file__), "golden", "hand_assembled_index_bin.py"))
    H = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(H)
    return H


@pytest.mark.parametrize("head", [20, 22, 23])
def test_hand_assembled_whole_index_bin(head):
    """the FILE WALK (version header, per-level header, length bytes, cumulative counts, segment head table, sorted key heads of
    20 / 22 / 23 bytes, re-based key bodies, an incomplete last level) on a file assembled by hand from the reference's writers
    (tests/golden/hand_assembled_index_bin.py: every field cites commit.rs / compress_postinglist.rs / index.rs) -- not by
    oracle/ref_format.py, the restated writer the other index.bin tests use"""
    H = _hand_index_bin()
    data, psum = H.build(head)
    ix = S.IndexBin(data, 1, head, H.SEGMENT_NUMBER_BITS)
    assert ix.indexed_doc_count == H.N_DOCS and ix.level_count == 2 and ix.positions_sum_normalized == psum
    assert ix.term_count == 3 and ix.ngram_keys_skipped == 0
    assert [int(k) for k in ix.term_keys] == sorted(H.EXPECT)  # term id = rank of the key hash
    for key, (docs, tfs) in H.EXPECT.items():
        d, f = ix.postings(ix.term_of_key(key))
        assert d.tolist() == list(docs) and f.tolist() == list(tfs), hex(key)
    ix.close()
    # a truncated file and a key head that points outside its segment are refused, not read past
    for bad in (data[:len(data) - 7], data[:4 + 2 + 65536 + 16 + 3]):
        with pytest.raises(Exception):
            S.IndexBin(bad, 1, head, H.SEGMENT_NUMBER_BITS)


@pytest.mark.gpu
def test_hand_assembled_index_bin_answers_like_its_arrays():
    """the image built from the hand-assembled file = the image built from the postings the file was assembled from: same lexical
    info (avgdl from the stored positions_sum_normalized / indexed_doc_count), same answers, and the oracle's scores"""
    from oracle import oracle as O
    H = _hand_index_bin()
    data, psum = H.build(20)
    ix = S.IndexBin(data, 1, 20, H.SEGMENT_NUMBER_BITS)
    a, b = S.Shard(0), S.Shard(0)
    a.upload_index_bin(ix)
    keys = sorted(H.EXPECT)
    dl = np.array(H.doclen_bytes(0) + H.doclen_bytes(1)[:H.LEVEL1_DOCS], np.uint8)
    offs = np.zeros(len(keys) + 1, np.uint64)
    offs[1:] = np.cumsum([len(H.EXPECT[k][0]) for k in keys])
    alld = np.concatenate([np.asarray(H.EXPECT[k][0], np.uint32) for k in keys])
    allt = np.concatenate([np.asarray(H.EXPECT[k][1], np.uint16) for k in keys])
    b.upload_lexical(H.N_DOCS, dl, offs, alld, allt)
    osh = O.Shard(H.N_DOCS, dl, offs, alld, allt)
    assert a.lexical_info() == b.lexical_info()
    for qt, op, q in ((S.QueryType.Union, O.OP_OR, [0, 1, 2]), (S.QueryType.Union, O.OP_OR, [0]), (S.QueryType.Intersection, O.OP_AND, [1, 2])):
        ra = a.search_lexical_batch(a.make_queries([q], qt), 10)
        rb_ = b.search_lexical_batch(b.make_queries([q], qt), 10)
        for x, y in zip(ra, rb_):
            assert np.array_equal(x, y)
        od, os_, otot = osh.search_exhaustive(q, op, 10)
        assert int(ra[3][0]) == otot and np.allclose(ra[1][0][:ra[2][0]], os_, rtol=1e-4)
    a.close()
    b.close()


def test_index_bin_tiers_keep_every_key():
    """ss_index_bin_tier: the frequent keys first (hash order), the rare keys behind them (hash order) -- nothing is dropped, a key
    is found with one binary search per tier, n-gram components stay consecutive"""
    rng = np.random.default_rng(5)
    n_docs = 100_000
    dl, terms = _corpus(rng, n_docs, [30_000, 12, 4_000, 3, 900, 20_000, 1, 55])
    ngram = _ngram_terms(rng, n_docs, 23)
    data = RF.write_index_bin(n_docs, dl, terms, rng, segment_number_bits=4, key_head_size=23, ngram_terms=ngram)
    ix = S.IndexBin(data, 1, 23, 4)
    before = {int(k): ix.postings(t) for t, k in enumerate(ix.term_keys) if ix.term_component[t] == 0}
    n_all = ix.term_count
    nd = ix.tier(1000)
    assert ix.term_count == n_all and 0 < nd < n_all
    keys = [int(k) for k in ix.term_keys]
    assert keys[:nd] == sorted(keys[:nd]) and keys[nd:] == sorted(keys[nd:])
    for t in range(n_all):
        n = len(ix.postings(t)[0])
        assert (n >= 1000) == (t < nd)
    for key, (d, f) in before.items():
        t = ix.term_of_key(key)
        d2, f2 = ix.postings(t)
        assert np.array_equal(d, d2) and np.array_equal(f, f2)
        for c in range(int(ix.term_components[t])):  # the components of an n-gram key follow one another inside their tier
            assert int(ix.term_keys[t + c]) == key and int(ix.term_component[t + c]) == c
    ix.close()


@pytest.mark.gpu
def test_tiered_index_bin_upload_answers_like_the_arrays():
    """an index.bin uploaded in two tiers (frequent keys -> dense image, rare keys -> sparse tier): queries over keys of both tiers
    answer like the same postings uploaded as one dense image, and like the oracle"""
    from oracle import oracle as O
    rng = np.random.default_rng(31)
    n_docs = 120_000
    sizes = [50_000, 9_000, 40, 700, 25_000, 3, 150, 1]
    dl, terms = _corpus(rng, n_docs, sizes)
    data = RF.write_index_bin(n_docs, dl, terms, rng)
    ix = S.IndexBin(data)
    nd = ix.tier(1000)
    assert nd == 3
    a, b = S.Shard(0), S.Shard(0)
    a.upload_index_bin(ix)
    assert a.sparse_info()[0] == len(sizes) - nd
    by_id = [terms[[int(t[0]) for t in terms].index(int(k))] for k in ix.term_keys]  # the file's terms in tiered id order
    offs = np.zeros(len(by_id) + 1, np.uint64)
    offs[1:] = np.cumsum([len(t[1]) for t in by_id])
    alld, allt = np.concatenate([t[1] for t in by_id]), np.concatenate([t[2] for t in by_id])
    b.upload_lexical(n_docs, dl, offs, alld, allt)
    osh = O.Shard(n_docs, dl, offs, alld, allt)
    assert [int(x) for x in a.posting_count(list(range(len(by_id))))] == [len(t[1]) for t in by_id]
    for qt, op, q in ((S.QueryType.Union, O.OP_OR, [0, 1, 4]), (S.QueryType.Union, O.OP_OR, [3, 6]), (S.QueryType.Intersection, O.OP_AND, [0, 3]),
                      (S.QueryType.Union, O.OP_OR, [5, 2, 7, 1]), (S.QueryType.Intersection, O.OP_AND, [4, 6, 0])):
        ra = a.search_lexical_batch(a.make_queries([q], qt), 10, reference_shortcuts=False)
        rb_ = b.search_lexical_batch(b.make_queries([q], qt), 10, reference_shortcuts=False)
        assert np.array_equal(ra[2], rb_[2]) and np.array_equal(ra[3], rb_[3]) and np.allclose(ra[1], rb_[1], rtol=1e-6)
        od, os_, otot = osh.search_exhaustive(q, op, 10)
        assert int(ra[3][0]) == otot and np.allclose(ra[1][0][:ra[2][0]], os_, rtol=1e-4)
    a.close()
    b.close()


@pytest.mark.gpu
def test_tiered_upload_of_an_index_bin_with_several_fields():
    """a multi-field index.bin in two tiers: the rare keys' merged lists in the sparse tier -- answers like the same entries uploaded
    as one dense multi-field image, and like the BM25F oracle"""
    from oracle import oracle as O
    rng = np.random.default_rng(43)
    n_docs, n_fields, longest = 90_000, 3, 1
    dl = np.stack([O.lex_doclen(n_docs, seed=O.LEX_SEED + 5 * f) for f in range(n_fields)])
    keys = sorted(int(k) & ~7 for k in rng.integers(1 << 40, 1 << 63, size=7, dtype=np.int64))
    terms = []
    for key, df in zip(keys, (20_000, 300, 6_000, 40, 11_000, 900, 2)):
        d, f, t = [], [], []
        for doc in np.sort(rng.choice(n_docs, size=df, replace=False)):
            fs = np.sort(rng.choice(n_fields, size=int(rng.integers(1, n_fields + 1)), replace=False)) if rng.random() < 0.6 else [longest]
            for x in fs:
                d.append(int(doc)); f.append(int(x)); t.append(int(min(rng.geometric(0.5), 30)))
        terms.append((key, np.array(d), np.array(f), np.array(t)))
    data = RF.write_index_bin(n_docs, dl, terms, rng, n_fields=n_fields, longest_field_id=longest)
    ix = S.IndexBin(data, n_fields)
    nd = ix.tier(2000)
    assert nd == 3
    boost = [1.5, 1.0, 0.5]
    a, b = S.Shard(0), S.Shard(0)
    a.upload_index_bin(ix, boost)
    assert a.sparse_info()[0] == len(terms) - nd
    by_id = [terms[keys.index(int(k))] for k in ix.term_keys]  # the file's terms in tiered id order
    offs = np.zeros(len(by_id) + 1, np.uint64)
    offs[1:] = np.cumsum([len(t[1]) for t in by_id])
    D, F, T = (np.concatenate([t[i] for t in by_id]) for i in (1, 2, 3))
    b.upload_lexical_fields(n_docs, dl, boost, offs, D.astype(np.uint32), F.astype(np.uint8), T.astype(np.uint16))
    n = len(by_id)
    assert [int(x) for x in a.posting_count(list(range(n)))] == [int(x) for x in b.posting_count(list(range(n)))]
    for qt, oop in ((S.QueryType.Union, O.OP_OR), (S.QueryType.Intersection, O.OP_AND)):
        for q in ([0, 3], [4], [1, 5, 2], [0, 1, 2], [6, 3], [5, 4, 0]):
            ra = a.search_lexical_batch(a.make_queries([q], qt), 10, reference_shortcuts=False)
            rb_ = b.search_lexical_batch(b.make_queries([q], qt), 10, reference_shortcuts=False)
            assert np.array_equal(ra[2], rb_[2]) and np.array_equal(ra[3], rb_[3]) and np.allclose(ra[1], rb_[1], rtol=1e-6)
            od, os_, otot, _ = O.search_fields_exhaustive(n_docs, dl, boost, offs, D, F, T, q, oop, 10)
            assert int(ra[3][0]) == otot and np.allclose(ra[1][0][:ra[2][0]], os_, rtol=1e-4)
    a.close()
    b.close()


def _python_writer_of(T, RF, O, key_head_size, positions_limit):
    """the same corpus through oracle/ref_format.py: single terms, then the n-gram keys with their component tfs / df bytes"""
    terms, ngt = [], []
    lut = lambda df: int(O.lib().so_int_to_byte4(int(df)))
    for k in range(T.n_keys):
        if T.key_df(k) == 0:
            continue
        docs, tfs, cnt, pos = T.key_postings(k)
        per = [p.tolist() for p in np.split(pos, np.cumsum(cnt.astype(np.int64))[:-1])]
        h = T.key_hash(k)
        if k < T.vocab:
            terms.append((h, docs.astype(np.int64), tfs.astype(np.int64), per))
        elif (2 if (h & 7) == 1 else 3) <= key_head_size - 20:
            ngt.append((h, docs.astype(np.int64), cnt.astype(np.int64), k, per))
    return terms, ngt, lut


@pytest.mark.parametrize("head,limit,n_fields,longest", [(23, 32768, 3, 1), (22, 300, 2, 0), (20, 32768, 3, 2)])
def test_c_indexer_multi_field_writes_what_the_restated_python_writer_writes(head, limit, n_fields, longest):
    """... with SEVERAL indexed fields (the docs' tokens cut into consecutive spans; field vectors in front of the records, embedded
    pointers with field tags, n-gram records with their components' field vectors): the same bytes as oracle/ref_format.py; the
    product's walker and multi-field decoder read every key back"""
    from oracle import oracle as O, textindex as TI
    T = TI.TextCorpus(11, 30_000, 2500, n_frequent=12, mean_len=12.0, topic_share=0.4, n_fields=n_fields, longest_field=longest)
    assert T.n_ngram_keys > 100 and T.doclen_fields.shape == (n_fields, T.n_docs)
    data = T.write_index_bin(key_head_size=head, positions_limit=limit)
    lut = lambda df: int(O.lib().so_int_to_byte4(int(df)))
    terms, ng_terms = [], []
    for k in range(T.n_keys):
        if T.key_df(k) == 0:
            continue
        h = T.key_hash(k)
        docs, flds, tfs, cnt, pos = T.key_entries(k, 0)
        own = cnt > 0
        per = [p.tolist() for p in np.split(pos, np.cumsum(cnt[own].astype(np.int64))[:-1])] if own.any() else []
        if k < T.vocab:
            assert own.all() and np.array_equal(tfs, cnt)
            terms.append((h, docs.astype(np.int64), flds.astype(np.int64), tfs.astype(np.int64), per))
            continue
        nc = 2 if (h & 7) == 1 else 3
        if nc > head - 20:
            continue
        vecs = {}
        for c in range(nc):
            cd, cf, ct, _, _ = T.key_entries(k, c, positions=False)
            for d_, f_, t_ in zip(cd.tolist(), cf.tolist(), ct.tolist()):
                vecs.setdefault(d_, [[] for _ in range(nc)])[c].append((f_, t_))
        d0, f0 = int(docs[own][0]), int(flds[own][0])
        toks = T.doc_field_tokens(d0, f0)
        ranks = [int(toks[per[0][0] + i]) for i in range(nc)]
        assert T.ngram_key(ranks) == k
        ng_terms.append((h, docs[own].astype(np.int64), flds[own].astype(np.int64), cnt[own].astype(np.int64), vecs, [lut(T.key_df(r)) for r in ranks], per))
    ref = RF.write_index_bin(T.n_docs, T.doclen_fields, terms, np.random.default_rng(0), key_head_size=head, positions_limit=limit, n_fields=n_fields,
                             longest_field_id=longest, ngram_terms=ng_terms)
    assert len(data) == len(ref)
    assert data == ref
    ix = S.IndexBin(data, n_fields, key_head_size=head)
    assert ix.indexed_doc_count == T.n_docs and ix.term_count >= len(terms)
    ix.close()


@pytest.mark.parametrize("head,limit", [(23, 32768), (22, 300), (20, 32768)])
def test_c_indexer_writes_what_the_restated_python_writer_writes(head, limit):
    """oracle/ss_textindex.c (the mini indexer that produces config-size index.bin files) against oracle/ref_format.py (the restated
    writer the hand-assembled fixtures pin): the same corpus -> the same bytes; and the product's walker reads every key back"""
    from oracle import oracle as O, textindex as TI
    T = TI.TextCorpus(7, 70_000, 3000, n_frequent=12, mean_len=7.0, topic_share=0.4)
    assert T.n_ngram_keys > 100 and T.n_tokens > 400_000
    data = T.write_index_bin(key_head_size=head, positions_limit=limit)
    terms, ngt, lut = _python_writer_of(T, RF, O, head, limit)
    ng_terms = []
    for h, docs, cnt, k, per in ngt:
        nc = 2 if (h & 7) == 1 else 3
        comp = np.stack([T.key_postings(k, c, False)[1] for c in range(nc)], 1).astype(np.int64)
        # the component ranks: the single terms whose tf in the key's first doc equals the component tfs are not unique -- take them
        # from the doc's tokens at the key's first position
        toks = T.doc_tokens(int(docs[0]))
        ranks = [int(toks[per[0][0] + i]) for i in range(nc)]
        assert T.ngram_key(ranks) == k
        ng_terms.append((h, docs, cnt, comp, [lut(T.key_df(r)) for r in ranks], per))
    ref = RF.write_index_bin(T.n_docs, T.doclen, terms, np.random.default_rng(0), key_head_size=head, positions_limit=limit, ngram_terms=ng_terms)
    assert len(data) == len(ref)
    assert data == ref
    ix = S.IndexBin(data, key_head_size=head)
    assert ix.indexed_doc_count == T.n_docs and ix.term_count >= len(terms)
    for h, docs, tfs, per in terms[::97]:
        d, t = ix.postings(ix.term_of_key(h))
        assert np.array_equal(d, docs) and np.array_equal(t, tfs)


@pytest.mark.parametrize("n_fields,longest,nc,limit", [(3, 1, 2, 32768), (2, 0, 3, 32768), (4, 3, 2, 500)])
def test_multi_field_ngram_key_positions_roundtrip(n_fields, longest, nc, limit):
    """an n-gram key's record in a multi-field index: the components' field vectors, then the key's own vector and positions
    (index_posting.rs:666-741) -- restated writer -> decoder: the key's positions come out behind component 0's entries, field by field"""
    rng = np.random.default_rng(33)
    n = 1500
    docs = np.sort(rng.choice(65536, size=n, replace=False))
    postings, vecs, e_docs, e_fields, e_counts, e_pos = [], {}, [], [], [], []
    for d in docs:
        comp_fields = sorted(rng.choice(n_fields, size=int(rng.integers(1, n_fields + 1)), replace=False).tolist())
        own_fields = sorted(rng.choice(comp_fields, size=int(rng.integers(1, len(comp_fields) + 1)), replace=False).tolist())
        vecs[int(d)] = [[(f, int(rng.integers(1, 300))) for f in comp_fields]] + \
                       [[(f, int(rng.integers(1, 40))) for f in sorted(rng.choice(n_fields, size=int(rng.integers(1, n_fields + 1)), replace=False).tolist())]
                        for _ in range(nc - 1)]
        for f in own_fields:
            ps = RF.random_positions(rng, int(rng.integers(1, 6)), 200)
            e_docs.append(int(d)); e_fields.append(f); e_counts.append(len(ps)); e_pos.append(ps)
    blk = RF.encode_term_fields(e_docs, e_fields, e_counts, n_fields, longest, rng, positions_limit=limit, ngram_vecs=vecs, positions=e_pos)[0]
    bid, ctp, cnt, pivot, body = blk
    assert cnt == n and (limit == 32768 or pivot < cnt)
    buf = np.frombuffer(body, np.uint8).copy()
    rb = N.RefBlock(bid, ctp, cnt - 1, pivot, buf.ctypes.data, len(buf))
    d16 = np.zeros(65536, np.uint16); first = np.zeros(65537, np.uint32)
    f8 = np.zeros(65536 * n_fields, np.uint8); t16 = np.zeros(65536 * n_fields, np.uint16); np16 = np.zeros(65536 * n_fields, np.uint16)
    npos = C.c_uint64()
    N.lib().ss_ref_decode_block_fields_ngram_positions(C.byref(rb), n_fields, longest, nc, N.ptr(d16, N.u16p), N.ptr(first, N.u32p), N.ptr(f8, N.u8p),
                                                      N.ptr(t16, N.u16p), N.ptr(np16, N.u16p), None, 0, C.byref(npos))
    pos = np.zeros(max(npos.value, 1), np.uint16)
    got = N.lib().ss_ref_decode_block_fields_ngram_positions(C.byref(rb), n_fields, longest, nc, N.ptr(d16, N.u16p), N.ptr(first, N.u32p), N.ptr(f8, N.u8p),
                                                            N.ptr(t16, N.u16p), N.ptr(np16, N.u16p), N.ptr(pos, N.u16p), len(pos), C.byref(npos))
    assert got == n and np.array_equal(d16[:n], docs)
    own = {}
    for d, f, ps in zip(e_docs, e_fields, e_pos):
        own.setdefault(d, {})[f] = ps
    want_pos, at = [], 0
    for i, d in enumerate(docs):
        ents = vecs[int(d)][0]
        assert first[i + 1] - first[i] == len(ents)
        for j, (f, tf) in enumerate(ents):
            e = first[i] + j
            ps = own[int(d)].get(f, [])
            assert f8[e] == f and t16[e] == tf and np16[e] == len(ps), (i, j)
            want_pos += ps
    assert pos[:npos.value].tolist() == want_pos


@pytest.mark.parametrize("tiered", [False, True])
def test_decode_all_equals_the_mini_indexers_lists(tiered):
    """ss_index_bin_decode_all -- the decoder of the image builders (terms in parallel on the worker threads, postings written straight
    into their final places, positions gathered per chunk) -- against the lists the mini indexer wrote the file from: every key's docs,
    the tf of every component, the key's own positions behind its first component; before and after ss_index_bin_tier.  Host only."""
    from oracle import textindex as TI
    T = TI.TextCorpus(9, 90_000, 2500, n_frequent=12, mean_len=9.0, topic_share=0.4)
    assert T.n_ngram_keys > 100
    ix = S.IndexBin(T.write_index_bin(key_head_size=23), key_head_size=23)
    if tiered:
        assert 0 < ix.tier(400) < ix.term_count
    by_hash = {}
    for k in range(T.n_keys):
        if T.key_df(k):
            by_hash[T.key_hash(k)] = k
    offs, docs, tfs, npos, pos = ix.decode_all(positions=True)
    offs2, docs2, tfs2 = ix.decode_all()
    assert np.array_equal(offs, offs2) and np.array_equal(docs, docs2) and np.array_equal(tfs, tfs2)
    assert len(offs) == ix.term_count + 1 and int(offs[-1]) == len(docs) == len(npos) and int(npos.astype(np.int64).sum()) == len(pos)
    pst = np.zeros(len(npos) + 1, np.int64)
    pst[1:] = np.cumsum(npos.astype(np.int64))
    seen = 0
    for t in range(ix.term_count):
        k = by_hash[int(ix.term_keys[t])]
        c = int(ix.term_component[t])
        d, tf, cnt, ps = T.key_postings(k, c, positions=(c == 0))
        a, b = int(offs[t]), int(offs[t + 1])
        assert np.array_equal(docs[a:b], d) and np.array_equal(tfs[a:b], tf), (t, k, c)
        if c == 0:
            assert np.array_equal(npos[a:b], cnt) and np.array_equal(pos[pst[a]:pst[b]], ps), (t, k)
        else:
            assert not npos[a:b].any(), (t, k, c)
        seen += 1
    assert seen == ix.term_count
    ix.close()
