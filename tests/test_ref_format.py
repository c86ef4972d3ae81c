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
