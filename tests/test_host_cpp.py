"""The C++ host mirror of the reference's search interface (seekstorm_amd/host): planner, per-shard seams, merge and
the batch coalescer, driven through the flat C shim host_capi.cpp.  CPU part: library loads, scalar pieces equal the
oracle, a host without a GPU degrades to empty results with the C-ABI error kept.  GPU part (-m gpu): Index::search
over two shards (lexical / vector / hybrid) against the oracle's merge, and coalesced concurrent vector searches."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_LIB = os.path.join(ROOT, "seekstorm_amd", "lib", "libseekstorm_host.so")
REL = 1e-4

u8p, u16p, u32p, u64p, f32p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint16, C.c_uint32, C.c_uint64, C.c_float))


def P(a, t):
    return None if a is None else a.ctypes.data_as(t)


@pytest.fixture(scope="module")
def H():
    from seekstorm_amd import _native as N
    N.lib()  # loads torch's HIP runtime first, then libseekstorm_hip.so (one runtime per process)
    if not os.path.exists(HOST_LIB):
        from seekstorm_amd import build as B
        B.build_host()
    L = C.CDLL(HOST_LIB)
    L.ssh_idf.restype = C.c_float
    L.ssh_idf.argtypes = [C.c_uint64, C.c_uint64]
    L.ssh_normalize_f32.argtypes = [f32p, C.c_uint64]
    L.ssh_threshold_raw.restype = C.c_float
    L.ssh_threshold_raw.argtypes = [C.c_float]
    L.ssh_vector_score.restype = C.c_float
    L.ssh_vector_score.argtypes = [C.c_float]
    L.ssh_index_create.restype = C.c_void_p
    L.ssh_index_create.argtypes = [C.c_int, C.POINTER(C.c_int)]
    L.ssh_index_destroy.argtypes = [C.c_void_p]
    L.ssh_shard_ok.argtypes = [C.c_void_p, C.c_int]
    L.ssh_shard_create_error.argtypes = [C.c_void_p, C.c_int]
    L.ssh_upload_lexical.argtypes = [C.c_void_p, C.c_int, C.c_uint64, u8p, C.c_uint32, u64p, u32p, u16p]
    L.ssh_upload_vectors.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, f32p, u32p]
    L.ssh_search.argtypes = [C.c_void_p, u32p, C.c_uint32, f32p, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32,
                             C.c_int, C.c_float, C.c_int, C.c_uint32, u64p, f32p, u8p, f32p, f32p, u64p]
    L.ssh_search_lexical_shard.argtypes = [C.c_void_p, C.c_int, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                           C.c_uint32, C.c_uint32, u64p, f32p, u64p]
    L.ssh_search_lexical_shard_ex.argtypes = [C.c_void_p, C.c_int, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.c_uint32, C.c_void_p, u32p, C.c_uint32, C.POINTER(C.c_uint16), C.c_uint32, C.c_void_p,
                                              C.c_uint32, C.c_uint32, u64p, f32p, u64p]
    L.ssh_index_search_sorted.argtypes = [C.c_void_p, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                          C.c_uint32, C.c_uint32, u64p, f32p, u64p]
    L.ssh_upload_lexical_fields.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, u8p, f32p, C.c_uint32, u64p, u32p, u8p, u16p]
    L.ssh_upload_lexical_fields_positions.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, u8p, f32p, C.c_uint32, u64p, u32p, u8p, u16p,
                                                      u16p, C.c_uint64]
    L.ssh_set_deleted.argtypes = [C.c_void_p, C.c_int, u64p, C.c_uint64]
    L.ssh_commit_level.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, u8p, C.c_uint32, C.c_uint32, u64p, u32p, u16p, u16p, C.c_uint64]
    L.ssh_facet_count.argtypes = [C.c_void_p, C.c_int, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u64p, C.c_uint32,
                                  C.c_void_p, C.c_uint32, C.c_void_p, u64p, u64p]
    L.ssh_coalesced_vector_search.argtypes = [C.c_void_p, C.c_int, C.c_uint32, f32p, C.c_uint32, C.c_uint32, C.c_uint32,
                                              u64p, f32p, u32p]
    L.ssh_upload_vectors_i8.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, C.c_void_p, u32p]
    L.ssh_quantize_f32_to_i8.argtypes = [f32p, C.c_uint64, C.c_void_p]
    L.ssh_coalesced_lexical_search.argtypes = [C.c_void_p, C.c_int, C.c_uint32, u32p, u32p, u32p, u32p, C.c_uint32, C.c_uint32,
                                               C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u64p, f32p, u32p, u64p]
    L.ssh_open_index_bin.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, u64p, C.c_uint32]
    L.ssh_upload_facets.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, C.c_void_p]
    L.ssh_search_lexical_shard_filtered.argtypes = [C.c_void_p, C.c_int, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                    C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, u64p, f32p, u64p]
    L.ssh_set_clusters.argtypes = [C.c_void_p, C.c_int, C.c_uint32, u32p, C.c_uint32, u32p]
    L.ssh_search_vector_shard_ann.argtypes = [C.c_void_p, C.c_int, f32p, C.c_uint32, C.c_int, C.c_uint32, C.c_float, C.c_uint32,
                                              u64p, f32p, u64p, u64p]
    L.ssh_turboquant.restype = None
    L.ssh_turboquant.argtypes = [f32p, C.c_uint64, f32p, C.c_uint64, C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ssh_index_search_lexical_batch.argtypes = [C.c_void_p, C.c_uint32, u32p, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                                 u64p, f32p, u32p, u64p]
    return L


def _search(L, ix, terms, qv, qt, mode, offset, length, rt=2, thr=None, normalize=True):
    t = np.ascontiguousarray(terms, np.uint32)
    cap = max(length, 1)
    doc = np.zeros(cap, np.uint64); sc = np.zeros(cap, np.float32); src = np.zeros(cap, np.uint8)
    ls = np.zeros(cap, np.float32); vs = np.zeros(cap, np.float32); meta = np.zeros(4, np.uint64)
    q = None if qv is None else np.ascontiguousarray(qv, np.float32)
    n = L.ssh_search(ix, P(t, u32p), len(t), P(q, f32p), qt, mode, offset, length, rt, 0 if thr is None else 1,
                     0.0 if thr is None else float(thr), 1 if normalize else 0, cap, P(doc, u64p), P(sc, f32p), P(src, u8p),
                     P(ls, f32p), P(vs, f32p), P(meta, u64p))
    return doc[:n], sc[:n], src[:n], ls[:n], vs[:n], meta


# ------------------------------------------------------------------ CPU
def test_host_scalars_match_oracle(H):
    from oracle import oracle as O
    OL = O.lib()
    for N_, n in ((1_000_000, 1000), (10, 3), (40_000, 39_999), (7, 7)):
        assert H.ssh_idf(N_, n) == OL.so_idf(N_, n)
    assert abs(H.ssh_idf(1_000_000, 1000) - 6.9072566) < 1e-6  # SURVEY 8c KAT
    v = (np.sin(0.137 * np.arange(128)) / 2 + np.cos(0.013 * np.arange(128)) / 2).astype(np.float32)  # make_f32, vector_similarity.rs:3012
    a = v.copy()
    H.ssh_normalize_f32(P(a, f32p), len(a))
    assert np.allclose(a, O.normalize(v), rtol=0, atol=1e-7) and abs(float(np.dot(a, a)) - 1.0) < 1e-5
    q8 = np.zeros(9, np.int8)
    vq = np.array([0.5 / 127, 1.5 / 127, -0.5 / 127, -1.5 / 127, 2.0, -3.0, 0.3, 0.0, 126.4 / 127], np.float32)
    H.ssh_quantize_f32_to_i8(P(vq, f32p), 9, q8.ctypes.data)
    assert np.array_equal(q8, O.quantize_i8(vq))  # quantize_f32_to_i8, vector_similarity.rs:1226-1232
    assert H.ssh_threshold_raw(0.7) == np.float32((np.float32(0.7) * np.float32(2) - np.float32(1)) / (np.float32(1) / np.float32(16129)))
    assert abs(H.ssh_vector_score(16129.0) - 1.0) < 1e-6 and abs(H.ssh_vector_score(0.0) - 0.5) < 1e-7


def test_host_without_gpu_degrades_to_empty(H):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tests")
    ix = H.ssh_index_create(2, None)
    try:
        assert H.ssh_shard_ok(ix, 0) == 0 and H.ssh_shard_create_error(ix, 0) == -3  # SS_EDEVICE, no CPU fallback
        d, s, src, _, _, meta = _search(H, ix, [1, 2], np.ones(8, np.float32), 1, 2, 0, 10)
        assert len(d) == 0 and meta[0] == 0 and meta[1] == 0
        assert np.int64(meta[3]) == -3  # the C-ABI code that emptied the result is kept (search.rs:2461-2463 degrade)
    finally:
        H.ssh_index_destroy(ix)


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_cpp_index_two_shards_matches_oracle(H):
    from oracle import oracle as O
    n_docs, dim, S_n = 40_000, 64, 2
    voc = [3000, 3600, 4000]
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = O.lex_corpus(n_docs, voc)
    rows = O.vec_gen(O.VEC_SEED, 0, n_docs, dim)
    qv = O.vec_gen(O.VECQ_SEED, 0, 1, dim, normalize=False)[0] * 3.0
    dev = (C.c_int * S_n)(0, 0)
    ix = H.ssh_index_create(S_n, dev)
    oshards, orows = [], []
    try:
        for sid in range(S_n):  # doc g -> shard g % S, local id g // S (index.rs:5284)
            sel = np.arange(sid, n_docs, S_n)
            o2, d2, t2 = [0], [], []
            for t in range(len(voc)):
                d = docs[int(offs[t]):int(offs[t + 1])]
                f = tfs[int(offs[t]):int(offs[t + 1])]
                m = (d % S_n) == sid
                d2.append(d[m] // S_n); t2.append(f[m]); o2.append(o2[-1] + int(m.sum()))
            d2 = np.concatenate(d2).astype(np.uint32); t2 = np.concatenate(t2).astype(np.uint16)
            o2 = np.asarray(o2, np.uint64)
            dls = np.ascontiguousarray(dl[sel])
            r = np.ascontiguousarray(rows[sel])
            assert H.ssh_shard_ok(ix, sid) == 1
            assert H.ssh_upload_lexical(ix, sid, len(sel), P(dls, u8p), len(voc), P(o2, u64p), P(d2, u32p), P(t2, u16p)) == 0
            assert H.ssh_upload_vectors(ix, sid, len(sel), dim, P(r, f32p), None) == 0
            oshards.append(O.Shard(len(sel), dls, o2, d2, t2))
            orows.append(r)
        k = 20
        qn = O.normalize(qv)
        lex_d, lex_s, vec_d, vec_s, lex_tot = [], [], [], [], 0
        for sid in range(S_n):
            od, os_, otot = oshards[sid].search_exhaustive([0, 1, 2], O.OP_OR, k)
            lex_d += [int(x) * S_n + sid for x in od]; lex_s += list(os_); lex_tot += otot
            vd, vs, _, _ = O.vec_search(orows[sid], qn, k)
            vec_d += [int(x) * S_n + sid for x in vd]; vec_s += list(vs)
        for mode in (0, 1, 2):  # Lexical, Vector, Hybrid
            d, s, src, ls, vsc, meta = _search(H, ix, [0, 1, 2], qv, 1, mode, 0, k)
            od, os_, osrc = O.merge(mode, (lex_d, lex_s), (vec_d, vec_s), 0, k)
            assert len(d) == len(od) == k and meta[0] == k and np.int64(meta[3]) == 0
            assert np.allclose(s, os_, rtol=REL, atol=2e-6)
            band = abs(float(os_[-1])) * REL + 2e-6
            assert {int(x) for x, y in zip(d, s) if y > os_[-1] + band} == {int(x) for x, y in zip(od, os_) if y > os_[-1] + band}
            if mode == 0:
                assert meta[1] == lex_tot and np.all(src == 0) and np.allclose(ls, s)
            if mode == 1:
                assert meta[2] == n_docs and np.all(src == 1)  # AnnMode::All observes every record
                assert np.allclose(vsc, (s / np.float32(16129.0) + 1) / 2, rtol=1e-6)
            if mode == 2:
                assert set(src.tolist()) <= {0, 1, 2}
        # offset / length are applied after the merge (search.rs:2109-2119)
        d0, s0, *_ = _search(H, ix, [0, 1, 2], None, 1, 0, 0, 12)
        d1, s1, *_ = _search(H, ix, [0, 1, 2], None, 1, 0, 5, 7)
        assert np.array_equal(d0[5:12], d1) and np.array_equal(s0[5:12], s1)
        # the per-shard seam drains its own offset (search.rs:3585-3593)
        doc = np.zeros(10, np.uint64); sc = np.zeros(10, np.float32); meta = np.zeros(4, np.uint64)
        n = H.ssh_search_lexical_shard(ix, 0, P(np.array([1, 2], np.uint32), u32p), 2, 0, 3, 7, 2, 10, P(doc, u64p), P(sc, f32p), P(meta, u64p))
        od, os_, otot = oshards[0].search_exhaustive([1, 2], O.OP_AND, 10)
        assert n == min(7, max(0, len(od) - 3)) and meta[1] == otot
        assert np.allclose(sc[:n], os_[3:3 + n], rtol=REL)
    finally:
        H.ssh_index_destroy(ix)


@pytest.mark.gpu
def test_cpp_coalescer_batches_concurrent_queries(H):
    from oracle import oracle as O
    n_rows, dim, nq, k = 6000, 96, 40, 10
    rows = O.vec_gen(21, 0, n_rows, dim)
    qs = O.vec_gen(22, 0, nq, dim)
    ix = H.ssh_index_create(1, None)
    try:
        assert H.ssh_upload_vectors(ix, 0, n_rows, dim, P(rows, f32p), None) == 0
        doc = np.zeros((nq, k), np.uint64); sc = np.zeros((nq, k), np.float32); cnt = np.zeros(nq, np.uint32)
        batches = H.ssh_coalesced_vector_search(ix, 0, nq, P(qs, f32p), k, 64, 20000, P(doc, u64p), P(sc, f32p), P(cnt, u32p))
        assert 1 <= batches < nq  # concurrent callers share device passes
        for i in range(nq):
            od, os_, _, _ = O.vec_search(rows, qs[i], k)
            assert cnt[i] == k and np.allclose(sc[i], os_, rtol=REL, atol=2e-6)
            assert set(map(int, doc[i][:5])) <= set(map(int, od))
    finally:
        H.ssh_index_destroy(ix)


@pytest.mark.gpu
def test_cpp_lexical_coalescer_batches_concurrent_queries(H):
    """concurrent single-query search_lexical_shard callers (no batched search API in the reference) share
    device batches; mixed offsets, NOT terms, every answer equal to its own single search"""
    from oracle import oracle as O
    n_docs, voc = 120_000, [3000, 3400, 3700, 3900, 4050]
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = O.lex_corpus(n_docs, voc)
    osh = O.Shard(n_docs, dl, offs, docs, tfs)
    ix = H.ssh_index_create(1, None)
    try:
        assert H.ssh_upload_lexical(ix, 0, n_docs, P(dl, u8p), len(voc), P(offs, u64p), P(docs, u32p), P(tfs, u16p)) == 0
        rng = np.random.default_rng(4)
        nq, length = 48, 10
        tl = [[int(x) for x in rng.choice(5, int(rng.integers(1, 4)), replace=False)] for _ in range(nq)]
        nl = [[t for t in range(5) if t not in q][:int(rng.integers(0, 2))] for q in tl]
        terms = np.array([t for q in tl for t in q], np.uint32)
        toff = np.zeros(nq + 1, np.uint32); toff[1:] = np.cumsum([len(q) for q in tl])
        nots = np.array([t for q in nl for t in q] + [0], np.uint32)
        noff = np.zeros(nq + 1, np.uint32); noff[1:] = np.cumsum([len(q) for q in nl])
        for qt, oop in ((1, O.OP_OR), (0, O.OP_AND)):
            doc = np.zeros((nq, length), np.uint64); sc = np.zeros((nq, length), np.float32)
            cnt = np.zeros(nq, np.uint32); tot = np.zeros(nq, np.uint64)
            batches = H.ssh_coalesced_lexical_search(ix, 0, nq, P(terms, u32p), P(toff, u32p), P(nots, u32p), P(noff, u32p), qt, 0,
                                                     length, 2, 1024, 20000, P(doc, u64p), P(sc, f32p), P(cnt, u32p), P(tot, u64p))
            assert 1 <= batches < nq
            for i in range(nq):
                off = i % 3  # the shim gives query i the offset i % 3
                od, os_, otot = osh.search_exhaustive(tl[i], oop, off + length, not_terms=nl[i])
                assert int(tot[i]) == otot
                assert cnt[i] == max(0, min(length, len(od) - off))
                assert np.allclose(sc[i][:cnt[i]], os_[off:off + cnt[i]], rtol=REL)
    finally:
        H.ssh_index_destroy(ix)


@pytest.mark.gpu
def test_cpp_index_vector_search_on_i8_image(H):
    """Index::search in Vector mode over Precision::I8 shards: the f32 query is normalised, quantised like the records
    (search.rs:1464-1490) and scored with the exact integer dot"""
    from oracle import oracle as O
    n_rows, dim, k = 4000, 128, 15
    rows = O.quantize_i8(O.vec_gen(31, 0, n_rows, dim))
    qv = O.vec_gen(32, 0, 1, dim, normalize=False)[0] * 2.5
    ix = H.ssh_index_create(1, None)
    try:
        assert H.ssh_upload_vectors_i8(ix, 0, n_rows, dim, rows.ctypes.data, None) == 0
        d, s, src, ls, vsc, meta = _search(H, ix, [], qv, 1, 1, 0, k)
        od, os_, _, _ = O.vec_search_i8(rows, O.quantize_i8(O.normalize(qv)), k)
        assert len(d) == k and np.array_equal(s, os_) and int(d[0]) == int(od[0])
        assert np.allclose(vsc, (s / np.float32(16129.0) + 1) / 2, rtol=1e-6)
    finally:
        H.ssh_index_destroy(ix)


@pytest.mark.gpu
def test_cpp_shard_ann_modes(H):
    """Shard::search_vector_shard with the reference's AnnMode values (vector.rs:1300-1307) over an i8 image: the C++ mirror
    converts the normalised cluster threshold like TopK::new and reports observed_cluster_count"""
    from oracle import oracle as O
    from seekstorm_amd.search import threshold_raw
    rng = np.random.default_rng(5)
    dim, lc = 64, [6, 5]
    child = rng.integers(50, 300, 11).astype(np.uint32)
    rows32 = O.vec_gen(41, 0, int(child.sum()), dim)
    rows = O.quantize_i8(rows32)
    qv = rows32[17] + 0.05 * O.vec_gen(42, 0, 1, dim)[0]
    qv = (qv / np.linalg.norm(qv)).astype(np.float32)
    q8 = O.quantize_i8(qv)
    first = np.concatenate([[0], np.cumsum(child.astype(np.int64))[:-1]])
    med = rows[first].astype(np.int32) @ q8.astype(np.int32)
    t_norm = float((np.float32(np.sort(med)[-4]) / np.float32(16129.0) + np.float32(1.0)) / np.float32(2.0))
    ix = H.ssh_index_create(1, None)
    try:
        assert H.ssh_upload_vectors_i8(ix, 0, len(rows), dim, rows.ctypes.data, None) == 0
        lcn, ccn = np.asarray(lc, np.uint32), child
        assert H.ssh_set_clusters(ix, 0, len(lcn), P(lcn, u32p), len(ccn), P(ccn, u32p)) == 0
        k = 12
        for kind, n, t, kw in ((2, 2, 0.0, dict(n_probe=2)), (1, 0, t_norm, dict(cluster_threshold_raw=threshold_raw(t_norm))),
                               (3, 1, t_norm, dict(n_probe=1, cluster_threshold_raw=threshold_raw(t_norm))), (0, 0, 0.0, None)):
            doc = np.zeros(k, np.uint64); sc = np.zeros(k, np.float32); meta = np.zeros(4, np.uint64); ncl = C.c_uint64()
            nres = H.ssh_search_vector_shard_ann(ix, 0, P(qv, f32p), k, kind, n, t, k, P(doc, u64p), P(sc, f32p), P(meta, u64p),
                                                 C.byref(ncl))
            assert int(meta[3]) == 0
            if kw is None:
                od, os_, _, oobs = O.vec_search_i8(rows, q8, k)
                assert ncl.value == 11
            else:
                od, os_, _, oobs, oncl = O.vec_search_i8_ann(rows, q8, k, lc, child, **kw)
                assert ncl.value == oncl
            assert nres == len(od) and np.array_equal(sc[:nres], os_)
            assert int(meta[2]) == oobs  # observed_vector_count: the records of the visited clusters (vector.rs:421, 1510)
        # ... and with tombstones only the live ones, in AnnMode::All too
        gone = np.array([3, 40, 41, 500], np.uint64)
        assert H.ssh_set_deleted(ix, 0, P(gone, u64p), len(gone)) == 0
        for kind, n, kw in ((2, 2, dict(n_probe=2)), (0, 0, {})):
            meta = np.zeros(4, np.uint64)
            H.ssh_search_vector_shard_ann(ix, 0, P(qv, f32p), k, kind, n, 0.0, k, P(doc, u64p), P(sc, f32p), P(meta, u64p), None)
            oobs = (O.vec_search_i8_ann(rows, q8, k, lc, child, deleted=gone, **kw) if kw else O.vec_search_i8_ann(rows, q8, k, None, None, deleted=gone))[3]
            assert int(meta[3]) == 0 and int(meta[2]) == oobs
        assert H.ssh_set_deleted(ix, 0, None, 0) == 0
        # Nprobe(0) has no TopK slot to push into: refused, the mirror degrades to an empty result with the code kept
        meta = np.zeros(4, np.uint64)
        assert H.ssh_search_vector_shard_ann(ix, 0, P(qv, f32p), k, 2, 0, 0.0, k, P(doc, u64p), P(sc, f32p), P(meta, u64p), None) == 0
        assert int(meta[3]) != 0
    finally:
        H.ssh_index_destroy(ix)


@pytest.mark.gpu
def test_cpp_shard_opens_an_index_bin_with_ngram_keys(H):
    """Shard::open_index_bin on a default-style index (23-byte key heads, bigram + trigram keys): term ids in key order, an
    n-gram key as consecutive component ids, and make_query scoring the components with idf_ngram_i (search.rs:3231-3262)"""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from oracle import oracle as O, ref_format as RF
    from test_ref_format import _corpus, _ngram_terms
    import seekstorm_amd as S
    rng = np.random.default_rng(44)
    n_docs, head = 100_000, 23
    dl, terms = _corpus(rng, n_docs, [30_000, 4_000])
    ngram = _ngram_terms(rng, n_docs, head, dfs=(5_000, 20_000))
    data = np.frombuffer(RF.write_index_bin(n_docs, dl, terms, rng, key_head_size=head, ngram_terms=ngram), np.uint8).copy()
    pix = S.IndexBin(data.tobytes(), 1, head)   # the same walk through the Python mirror: lists for the oracle
    lists = [pix.postings(t) for t in range(pix.term_count)]
    offs = np.zeros(len(lists) + 1, np.uint64)
    offs[1:] = np.cumsum([len(l[0]) for l in lists])
    osh = O.Shard(n_docs, dl, offs, np.concatenate([l[0] for l in lists]), np.concatenate([l[1] for l in lists]))
    ix = H.ssh_index_create(1, None)
    try:
        keys = np.zeros(64, np.uint64)
        nt = H.ssh_open_index_bin(ix, 0, data.ctypes.data, len(data), head, P(keys, u64p), 64)
        assert nt == pix.term_count == 2 + 2 + 3 and np.array_equal(keys[:nt], pix.term_keys)
        tri = pix.terms_of_key(ngram[1][0])
        single = pix.term_of_key(terms[0][0])
        tl = [t for t, _ in tri] + [single]
        idf = [w for _, w in tri] + [float(S.idf_f32(n_docs, len(lists[single][0])))]
        for qt, op in ((1, O.OP_OR), (0, O.OP_AND)):  # SS_OP_UNION = 1, SS_OP_INTERSECTION = 0
            t = np.ascontiguousarray(tl, np.uint32)
            doc = np.zeros(10, np.uint64); sc = np.zeros(10, np.float32); meta = np.zeros(4, np.uint64)
            n = H.ssh_search_lexical_shard(ix, 0, P(t, u32p), len(t), qt, 0, 10, 2, 10, P(doc, u64p), P(sc, f32p), P(meta, u64p))
            od, os_, otot = osh.search_exhaustive(tl, op, 10, idf=idf)
            assert int(meta[3]) == 0 and n == len(od) and int(meta[1]) == otot
            assert np.allclose(sc[:n], os_, rtol=1e-4)
    finally:
        H.ssh_index_destroy(ix)
        pix.close()


@pytest.mark.gpu
def test_cpp_shard_facet_filter(H):
    """Shard::search_lexical_shard with a FacetFilter (numeric range + string set) against the oracle with the failing docs
    excluded; offset / length drain on top"""
    from oracle import oracle as O
    import seekstorm_amd as S
    n_docs = 80_000
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = O.lex_corpus(n_docs, [4095, 4000, 3000])  # three frequent vocabulary terms
    osh = O.Shard(n_docs, dl, offs, docs, tfs)
    rng = np.random.default_rng(3)
    rec = np.dtype([("x", "<i4"), ("s", "<u2")])
    v = np.zeros(n_docs, rec)
    v["x"] = rng.integers(-100, 100, n_docs); v["s"] = rng.integers(0, 12, n_docs)
    keep = (v["x"] >= -20) & (v["x"] < 35) & np.isin(v["s"], [2, 5, 11])
    osh.set_deleted(np.nonzero(~keep)[0])
    farr, nf = S.Shard.facet_filters([(0, "i32", -20, 35), (4, "string16", [2, 5, 11])])
    ix = H.ssh_index_create(1, None)
    try:
        assert H.ssh_upload_lexical(ix, 0, n_docs, P(dl, u8p), len(offs) - 1, P(offs, u64p), P(docs, u32p), P(tfs, u16p)) == 0
        raw = v.view(np.uint8).reshape(n_docs, rec.itemsize).copy()
        assert H.ssh_upload_facets(ix, 0, n_docs, rec.itemsize, raw.ctypes.data) == 0
        for qt, op in ((1, O.OP_OR), (0, O.OP_AND)):
            t = np.ascontiguousarray([0, 1], np.uint32)
            doc = np.zeros(10, np.uint64); sc = np.zeros(10, np.float32); meta = np.zeros(4, np.uint64)
            n = H.ssh_search_lexical_shard_filtered(ix, 0, P(t, u32p), 2, qt, 3, 10, 2, nf, C.cast(farr, C.c_void_p), 10,
                                                    P(doc, u64p), P(sc, f32p), P(meta, u64p))
            od, os_, otot = osh.search_exhaustive([0, 1], op, 13)
            assert int(meta[3]) == 0 and int(meta[1]) == otot and n == len(od) - 3
            assert np.allclose(sc[:n], os_[3:], rtol=1e-4) and all(keep[int(d)] for d in doc[:n])
    finally:
        H.ssh_index_destroy(ix)


def test_turboquant_quantiser_mirrors_equal_the_oracle(H):
    """Quantization::TurboQuantI8, query side (TurboQuant::quantize_f32_i8 and its AVX2 form, vector_similarity.rs:1927-1983):
    the C++ and the Python mirror against the oracle's restatement, bit for bit -- both summation orders, dims that are and are
    not powers of two, a zero vector (scale floor 1e-8)"""
    import seekstorm_amd as S
    from oracle import oracle as O
    rng = np.random.default_rng(8)
    for n in (5, 64, 100, 768):
        dim = S.turboquant_dim(n)
        assert dim == {5: 8, 64: 64, 100: 128, 768: 1024}[n]
        mask = np.where(rng.random(dim) < 0.5, 1.0, -1.0).astype(np.float32)
        rows = (rng.standard_normal((6, n)) * rng.choice([1e-3, 1.0, 40.0])).astype(np.float32)
        rows[5] = 0.0
        for avx2 in (False, True):
            oq, osc, onm = O.turboquant_i8(rows, mask, avx2)
            for i in range(len(rows)):
                q, sc, nm = S.turboquant_f32_to_i8(rows[i], mask, avx2)
                assert np.array_equal(q, oq[i]) and np.float32(sc) == osc[i] and np.float32(nm) == onm[i], (n, avx2, i)
                cq = np.zeros(dim, np.int8); cs, cn = C.c_float(), C.c_float()
                H.ssh_turboquant(P(rows[i], f32p), n, P(mask, f32p), dim, 1 if avx2 else 0, cq.ctypes.data, C.byref(cs), C.byref(cn))
                assert np.array_equal(cq, oq[i]) and np.float32(cs.value) == osc[i] and np.float32(cn.value) == onm[i], (n, avx2, i)
        assert osc[5] == np.float32(1e-8) and not oq[5].any()


@pytest.mark.gpu
def test_cpp_index_lexical_batch_host_gather_and_device_exchange(H):
    """Index::search_lexical_batch: (a) two shards on one GPU -> host gather, every query equal to Index::search's single-query
    answer; (b) a communicator needs one GPU per shard: two shards on device 0 -> SS_EINVAL; (c) one shard with the device
    exchange on (ss_bm25_search_sharded over a communicator of one) -> the same lists as the host path"""
    from oracle import oracle as O
    n_docs, S_n = 60_000, 2
    voc = [3000, 3300, 3600, 4000, 4040]
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = O.lex_corpus(n_docs, voc)
    queries = [[0, 1, 2], [3, 4], [2], [1, 3, 4], [0, 4]]
    tcat = np.asarray([t for q in queries for t in q], np.uint32)
    toff = np.asarray(np.cumsum([0] + [len(q) for q in queries]), np.uint32)
    k = 15

    def build(S):
        ix = H.ssh_index_create(S, (C.c_int * S)(*([0] * S)))
        for sid in range(S):
            o2, d2, t2 = [0], [], []
            for t in range(len(voc)):
                d = docs[int(offs[t]):int(offs[t + 1])]
                f = tfs[int(offs[t]):int(offs[t + 1])]
                m = (d % S) == sid
                d2.append(d[m] // S); t2.append(f[m]); o2.append(o2[-1] + int(m.sum()))
            d2 = np.concatenate(d2).astype(np.uint32); t2 = np.concatenate(t2).astype(np.uint16)
            dls = np.ascontiguousarray(dl[sid::S])
            assert H.ssh_upload_lexical(ix, sid, len(dls), P(dls, u8p), len(voc), P(np.asarray(o2, np.uint64), u64p), P(d2, u32p),
                                        P(t2, u16p)) == 0
        return ix

    def batch(ix, qt, exchange):
        doc = np.zeros((len(queries), k), np.uint64); sc = np.zeros((len(queries), k), np.float32)
        cnt = np.zeros(len(queries), np.uint32); tot = np.zeros(len(queries), np.uint64)
        rc = H.ssh_index_search_lexical_batch(ix, len(queries), P(tcat, u32p), P(toff, u32p), qt, k, 2, exchange, P(doc, u64p),
                                              P(sc, f32p), P(cnt, u32p), P(tot, u64p))
        return rc, doc, sc, cnt, tot

    ix2 = build(2)
    ix1 = build(1)
    try:
        for qt in (0, 1):  # Intersection, Union
            rc, doc, sc, cnt, tot = batch(ix2, qt, 0)
            assert rc == 0
            for i, q in enumerate(queries):
                d, s, src, ls, vs, meta = _search(H, ix2, q, None, qt, 0, 0, k)
                assert cnt[i] == len(d) and tot[i] == meta[1]
                assert np.array_equal(doc[i, :cnt[i]], d) and np.array_equal(sc[i, :cnt[i]], s)
            assert batch(ix2, qt, 1)[0] == -1  # SS_EINVAL: both shards on device 0
            rc_h, dh, sh_, ch, th = batch(ix1, qt, 0)
            assert rc_h == 0
        d_before = _search(H, ix1, queries[0], None, 1, 0, 2, 7)  # Index::search on the host path, offset 2
        for qt in (0, 1):
            rc_h, dh, sh_, ch, th = batch(ix1, qt, 0)
            rc_x, dx, sx, cx, tx = batch(ix1, qt, 1)
            assert rc_h == 0 and rc_x == 0
            assert np.array_equal(ch, cx) and np.array_equal(th, tx)
            for i in range(len(queries)):
                assert np.array_equal(dh[i, :ch[i]], dx[i, :cx[i]]) and np.array_equal(sh_[i, :ch[i]], sx[i, :cx[i]])
        # Index::search itself goes through the exchange once it is enabled: same answer, offset applied after the merge
        d_after = _search(H, ix1, queries[0], None, 1, 0, 2, 7)
        assert np.array_equal(d_before[0], d_after[0]) and np.array_equal(d_before[1], d_after[1]) and d_before[5][1] == d_after[5][1]
    finally:
        H.ssh_index_destroy(ix2)
        H.ssh_index_destroy(ix1)


class _SortC(C.Structure):  # ssh_result_sort
    _fields_ = [("facet_offset", C.c_uint32), ("facet_type", C.c_uint32), ("descending", C.c_uint32), ("reserved", C.c_uint32),
                ("base", C.c_double * 2)]


def _cpp_lexical_ex(H, ix, terms, qt, offset, length, rt, facet_filter=None, not_terms=(), field_filter=(), sorts=()):
    import seekstorm_amd as S
    from seekstorm_amd import _native as N
    t = np.ascontiguousarray(terms, np.uint32)
    nt = np.ascontiguousarray(list(not_terms), np.uint32)
    ff = np.ascontiguousarray(list(field_filter), np.uint16)
    farr, nf = S.Shard.facet_filters(facet_filter) if facet_filter else (None, 0)
    sarr = (_SortC * max(len(sorts), 1))()
    for i, so in enumerate(sorts):
        sarr[i].facet_offset, sarr[i].facet_type, sarr[i].descending = so[0], N.FACET_TYPES[so[1]], 1 if so[2] else 0
        if so[1] == "point":
            sarr[i].base[0], sarr[i].base[1] = so[3]
    cap = offset + length + 1
    doc = np.zeros(cap, np.uint64); sc = np.zeros(cap, np.float32); meta = np.zeros(4, np.uint64)
    n = H.ssh_search_lexical_shard_ex(ix, 0, P(t, u32p), len(t), qt, offset, length, rt, nf, None if farr is None else C.cast(farr, C.c_void_p),
                                      P(nt, u32p), len(nt), ff.ctypes.data_as(C.POINTER(C.c_uint16)), len(ff), C.cast(sarr, C.c_void_p),
                                      len(sorts), cap, P(doc, u64p), P(sc, f32p), P(meta, u64p))
    assert int(meta[3]) == 0, int(np.int64(meta[3]))
    return doc[:n].copy(), sc[:n].copy(), int(meta[1])


@pytest.mark.gpu
def test_cpp_shard_result_sort_point_facets_and_counts(H):
    """The C++ mirror's search_lexical_shard with result_sort (numeric and Point sort fields, a Point distance filter) and
    facet_count (numeric ranges, Point distance ranges) answer exactly like the Python mirror -- which the parity tests pin to
    the oracle (test_result_sort_by_facets, test_point_facets) -- on the same shard contents."""
    from oracle import oracle as O
    import seekstorm_amd as S
    n_docs = 120_000
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = O.lex_corpus(n_docs, [4095, 4000, 3000])
    rng = np.random.default_rng(29)
    rec = np.dtype([("c", "<i4"), ("a", "u1"), ("loc", "<u8"), ("g", "<f8")])
    v = np.zeros(n_docs, rec)
    v["c"] = rng.integers(-1000, 1000, n_docs); v["a"] = rng.integers(0, 4, n_docs); v["g"] = rng.standard_normal(n_docs)
    v["loc"] = O.morton_encode(np.round(rng.random(n_docs) * 50.0 + 10.0, 1), np.round(rng.random(n_docs) * 60.0 + 5.0, 1))
    raw = v.view(np.uint8).reshape(n_docs, rec.itemsize).copy()
    off = {n: rec.fields[n][1] for n in rec.names}
    psh = S.Shard(0)
    psh.upload_lexical(n_docs, dl, offs, docs, tfs)
    psh.upload_facets(raw)
    ix = H.ssh_index_create(1, None)
    try:
        assert H.ssh_upload_lexical(ix, 0, n_docs, P(dl, u8p), len(offs) - 1, P(offs, u64p), P(docs, u32p), P(tfs, u16p)) == 0
        assert H.ssh_upload_facets(ix, 0, n_docs, rec.itemsize, raw.ctypes.data) == 0
        base = (38.9, 30.2)
        pfilter = [(off["loc"], "point", base, 100.0, 1500.0, "km")]
        for terms, qt in (([0, 1, 2], 1), ([0, 1], 0)):
            q = psh.make_queries([terms], S.QueryType(qt))
            for sorts in ([(off["c"], "i32", True)], [(off["a"], "u8", False), (off["g"], "f64", True)], [(off["loc"], "point", False, base)],
                          [(off["a"], "u8", True), (off["loc"], "point", True, base), (off["c"], "i32", False)]):
                for k, flt in ((10, None), (33, pfilter), (25, [(off["c"], "i32", -500, 700)])):
                    pd, ps, ptot = psh.search_lexical_sorted(q, sorts, k, facet_filter=flt)
                    cd, cs, ctot = _cpp_lexical_ex(H, ix, terms, qt, 0, k, 2, facet_filter=flt, sorts=sorts)
                    assert ctot == ptot and len(cd) == len(pd), (terms, sorts, k)
                    assert np.allclose(cs, ps, rtol=1e-6), (terms, sorts, k)
                    for name in ("c", "a", "loc", "g"):
                        assert np.array_equal(v[name][cd.astype(np.int64)], v[name][pd]), (terms, sorts, k, name)
            # offset drains the head of the sorted list
            pd, ps, _ = psh.search_lexical_sorted(q, [(off["c"], "i32", True)], 20)
            cd, cs, _ = _cpp_lexical_ex(H, ix, terms, qt, 5, 15, 2, sorts=[(off["c"], "i32", True)])
            assert np.allclose(cs, ps[5:], rtol=1e-6)
            # facet counts
            t = np.ascontiguousarray(terms, np.uint32)
            for ftype, foff, bounds, pt in (("i32", off["c"], np.array([-500, 0, 250], np.int64).view(np.uint64), None),
                                            ("point", off["loc"], np.array([0.0, 500.0, 1000.0, 2000.0]).view(np.uint64), base)):
                out = np.zeros(len(bounds) + 1, np.uint64)
                tot = C.c_uint64()
                from seekstorm_amd import _native as N
                ptc = N.FacetPointC(pt[0], pt[1], 1, 0) if pt else None
                farr, nf = S.Shard.facet_filters(pfilter)
                rc = H.ssh_facet_count(ix, 0, P(t, u32p), len(t), qt, foff, N.FACET_TYPES[ftype], 0, P(bounds, u64p), len(bounds),
                                       C.byref(ptc) if ptc else None, nf, C.cast(farr, C.c_void_p), P(out, u64p), C.byref(tot))
                assert rc == 0
                if pt:
                    pc, pother, ptot = psh.facet_count(q, foff, "point", range_lower_bounds=bounds.view(np.float64), base=pt, unit="km", facet_filter=pfilter)
                else:
                    pc, pother, ptot = psh.facet_count(q, foff, ftype, range_lower_bounds=bounds.view(np.int64), facet_filter=pfilter)
                assert tot.value == ptot and np.array_equal(out[:-1], pc) and int(out[-1]) == pother
    finally:
        H.ssh_index_destroy(ix)
        psh.close()


@pytest.mark.gpu
def test_cpp_shard_union_under_a_field_filter(H):
    """The C++ mirror's search_lexical_shard for a union of several terms under a field filter (the reference's sub-query
    decomposition, union.rs:1168-1479) against the Python mirror, which
    test_union_under_a_field_filter_follows_the_reference_decomposition pins to a brute-force oracle of the rule; NOT terms on top"""
    from oracle import oracle as O
    import seekstorm_amd as S
    from test_gpu_parity import _fields_corpus
    n_docs, n_fields = 60_000, 3
    dfs = [30_000, 18_000, 5_000, 22_000]
    dl, offs, docs, fields, tfs = _fields_corpus(O, n_docs, n_fields, dfs, 41)
    boost = np.array([1.5, 1.0, 0.5], np.float32)
    psh = S.Shard(0)
    psh.upload_lexical_fields(n_docs, dl, boost, offs, docs, fields, tfs)
    ix = H.ssh_index_create(1, None)
    try:
        dlc = np.ascontiguousarray(dl, np.uint8)
        assert H.ssh_upload_lexical_fields(ix, 0, n_docs, n_fields, P(dlc.reshape(-1), u8p), P(boost, f32p), len(offs) - 1, P(offs, u64p),
                                           P(docs, u32p), P(np.ascontiguousarray(fields, np.uint8), u8p), P(tfs, u16p)) == 0
        for filt in ([0], [2], [0, 2]):
            for terms, neg in (([0, 1], []), ([0, 1, 3], []), ([2, 3], [0]), ([0, 1, 2, 3], [])):
                for rt in (S.ResultType.TopkCount, S.ResultType.Topk, S.ResultType.Count):
                    ro = psh.search_lexical_shard(terms, S.QueryType.Union, 2, 20, rt, strict=True, not_terms=neg, field_filter=filt)
                    cd, cs, ctot = _cpp_lexical_ex(H, ix, terms, 1, 2, 20, int(rt), not_terms=neg, field_filter=filt)
                    assert ctot == ro.result_count_total, (filt, terms, neg, rt)
                    # (the mirrors' idf may differ in the last place: logf here, numpy's float32 log there)
                    assert np.allclose(cs, np.array([r.score for r in ro.results], np.float32), rtol=1e-6), (filt, terms, neg, rt)
                    assert list(cd) == [r.doc_id for r in ro.results], (filt, terms, neg, rt)
                # an intersection under the filter goes straight to the kernels
                ro = psh.search_lexical_shard(terms, S.QueryType.Intersection, 0, 10, S.ResultType.TopkCount, strict=True, not_terms=neg, field_filter=filt)
                cd, cs, ctot = _cpp_lexical_ex(H, ix, terms, 0, 0, 10, 2, not_terms=neg, field_filter=filt)
                assert ctot == ro.result_count_total and np.allclose(cs, np.array([r.score for r in ro.results], np.float32), rtol=1e-6)
    finally:
        H.ssh_index_destroy(ix)
        psh.close()


@pytest.mark.gpu
def test_cpp_shard_commits_levels_on_a_tiered_vocabulary(H):
    """Shard::commit_level of the C++ mirror (commit.rs:142-148 as the seam sees it: a level's postings of all known terms, dense ids
    first, rare terms behind them) level by level, the last level half full first and re-committed whole, against the Python mirror's
    one-shot upload of the same docs (dense image + whole sparse lists): unions, intersections, NOT terms over both tiers"""
    from oracle import oracle as O
    import seekstorm_amd as S
    n_docs, nd = 150_000, 3
    rng = np.random.default_rng(77)
    dl = O.lex_doclen(n_docs)
    lists = []
    for df in (40_000, 15_000, 60_000, 5_000, 300, 20, 1_500):
        d = np.sort(rng.choice(n_docs, df, replace=False)).astype(np.uint32)
        lists.append((d, np.minimum(rng.geometric(0.5, df), 40).astype(np.uint16)))

    def csr(terms, d0, d1):
        o, dd, tt = [0], [], []
        for t in terms:
            d, tf = lists[t]
            i0, i1 = int(np.searchsorted(d, d0)), int(np.searchsorted(d, d1))
            dd.append(d[i0:i1]); tt.append(tf[i0:i1]); o.append(o[-1] + (i1 - i0))
        return np.asarray(o, np.uint64), np.concatenate(dd), np.concatenate(tt)

    ix = H.ssh_index_create(1, None)
    psh = None
    try:
        for l, d1 in [(0, 65536), (1, 131072), (2, 140_000), (2, n_docs)]:
            d0 = l << 16
            o, dd, tt = csr(range(len(lists)), d0, d1)
            lv = np.ascontiguousarray(dl[d0:d1], np.uint8)
            assert H.ssh_commit_level(ix, 0, l, d1 - d0, P(lv, u8p), len(lists), nd, P(o, u64p), P(dd, u32p), P(tt, u16p), None, 0) == 0
            if psh is not None:
                psh.close()
            psh = S.Shard(0)
            o, dd, tt = csr(range(nd), 0, d1)
            psh.upload_lexical(d1, dl[:d1], o, dd, tt)
            o, dd, tt = csr(range(nd, len(lists)), 0, d1)
            assert psh.append_sparse(o, dd, tt) == nd
            for terms, neg in (([0, 3], []), ([4, 1, 2], []), ([5, 6, 0], []), ([3, 4], []), ([6], []), ([0, 1], [3]), ([3, 6], [1]), ([1, 2], [])):
                for qt in (S.QueryType.Union, S.QueryType.Intersection):
                    ro = psh.search_lexical_shard(terms, qt, 0, 10, S.ResultType.TopkCount, strict=True, not_terms=neg)
                    cd, cs, ctot = _cpp_lexical_ex(H, ix, terms, int(qt), 0, 10, 2, not_terms=neg)
                    assert ctot == ro.result_count_total, (l, d1, terms, neg, qt)
                    assert np.allclose(cs, np.array([r.score for r in ro.results], np.float32), rtol=1e-6), (l, d1, terms, neg, qt)
                    assert list(cd) == [r.doc_id for r in ro.results], (l, d1, terms, neg, qt)
    finally:
        H.ssh_index_destroy(ix)
        if psh is not None:
            psh.close()


@pytest.mark.gpu
def test_cpp_shard_phrase_over_several_fields(H):
    """QueryType::Phrase through the C++ mirror on an image with several indexed fields (positions per (term, doc, field) entry;
    add_result.rs:3248-3386), with and without a field filter, against the oracle; and the multi-field all_terms_frequent marking"""
    from oracle import oracle as O
    from test_gpu_phrase import _corpus_fields
    n_docs, n_fields = 80_000, 3
    dfs = [45_000, 42_000, 9_000, 6_000]
    plant = [([0, 1], 0, 200), ([0, 1, 2], 1, 90), ([3, 2], 2, 50), ([1, 1], 2, 40)]
    dl, offs, docs, fields, tfs, positions = _corpus_fields(O, n_docs, n_fields, dfs, 29, plant, [(0, 1, 150)])
    boost = np.array([1.5, 1.0, 0.5], np.float32)
    ix = H.ssh_index_create(1, None)
    try:
        dlc = np.ascontiguousarray(dl, np.uint8)
        assert H.ssh_upload_lexical_fields_positions(ix, 0, n_docs, n_fields, P(dlc.reshape(-1), u8p), P(boost, f32p), len(offs) - 1,
                                                     P(offs, u64p), P(docs, u32p), P(fields, u8p), P(tfs, u16p), P(positions, u16p),
                                                     len(positions)) == 0
        for filt in ((), (0,), (1, 2)):
            for ph in ([0, 1], [0, 1, 2], [3, 2], [1, 1], [1, 0]):
                uniq = list(dict.fromkeys(ph))
                od, os_, otot = O.search_fields_phrase(n_docs, dl, boost, offs, docs, fields, tfs, positions, uniq, [uniq.index(w) for w in ph], 10,
                                                       field_filter=filt)
                cd, cs, ctot = _cpp_lexical_ex(H, ix, ph, 2, 0, 10, 2, field_filter=filt)  # QueryType::Phrase = 2, TopkCount
                assert ctot == otot, (filt, ph, ctot, otot)
                assert len(cd) == len(od) and np.allclose(cs, os_, rtol=REL)
        # terms 0 and 1 are in more than half of the docs and 80 000 > 256 * 10: the mirror marks the intersection like the reference
        od, os_, otot = O.search_fields_shortcut(n_docs, dl, boost, offs, docs, fields, tfs, [0, 1], 10)
        plain = O.search_fields_exhaustive(n_docs, dl, boost, offs, docs, fields, tfs, [0, 1], O.OP_AND, 10)
        cd, cs, ctot = _cpp_lexical_ex(H, ix, [0, 1], 0, 0, 10, 2)  # QueryType::Intersection = 0
        assert ctot == otot == plain[2] and len(cd) == len(od) and np.allclose(cs, os_, rtol=REL)
        assert not np.array_equal(plain[0], od)
    finally:
        H.ssh_index_destroy(ix)


@pytest.mark.gpu
def test_cpp_index_sorted_search_over_two_shards(H):
    """Index::search_lexical_sorted (result_ordering_root, min_heap.rs:56-300: the shards' sorted lists merged under the same
    order, each doc's facet values read from its own shard) against the Python mirror's Index.search(result_sort=...) on the
    same two shards' contents: numeric and Point sort fields, a facet filter, offset / length"""
    from oracle import oracle as O
    import seekstorm_amd as S
    from seekstorm_amd import _native as N
    n_docs, S_n = 50_000, 2
    voc = [3000, 3600, 4000]
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = O.lex_corpus(n_docs, voc)
    rng = np.random.default_rng(8)
    rec = np.dtype([("u", "<u4"), ("a", "u1"), ("loc", "<u8")])
    v = np.zeros(n_docs, rec)
    v["u"] = rng.permutation(n_docs); v["a"] = rng.integers(0, 5, n_docs)
    v["loc"] = O.morton_encode(rng.random(n_docs) * 50 + 10, rng.random(n_docs) * 60 + 5)
    raw = v.view(np.uint8).reshape(n_docs, rec.itemsize)
    ix = H.ssh_index_create(S_n, None)
    pshards = []
    try:
        for sid in range(S_n):
            sel = np.arange(sid, n_docs, S_n)
            o2, d2, t2 = [0], [], []
            for t in range(len(voc)):
                d = docs[int(offs[t]):int(offs[t + 1])]
                f = tfs[int(offs[t]):int(offs[t + 1])]
                m = (d % S_n) == sid
                d2.append(d[m] // S_n); t2.append(f[m]); o2.append(o2[-1] + int(m.sum()))
            o2 = np.asarray(o2, np.uint64); d2 = np.concatenate(d2).astype(np.uint32); t2 = np.concatenate(t2)
            dls = np.ascontiguousarray(dl[sel]); rs_ = np.ascontiguousarray(raw[sel])
            assert H.ssh_upload_lexical(ix, sid, len(sel), P(dls, u8p), len(voc), P(o2, u64p), P(d2, u32p), P(t2, u16p)) == 0
            assert H.ssh_upload_facets(ix, sid, len(sel), rec.itemsize, rs_.ctypes.data) == 0
            sh = S.Shard(0, shard_id=sid)
            sh.upload_lexical(len(sel), dls, o2, d2, t2)
            sh.upload_facets(rs_)
            pshards.append(sh)
        pidx = S.Index(pshards)
        base = (38.9, 30.2)
        for terms, qt in (([0, 1, 2], 1), ([0, 1], 0)):
            for sorts in ([(0, "u32", True)], [(4, "u8", False), (0, "u32", False)], [(5, "point", False, base)]):
                for off_, length, flt in ((0, 20, None), (6, 12, [(0, "u32", 4000, 45000)])):
                    ro = pidx.search(terms, None, S.QueryType(qt), S.SearchMode.Lexical, off_, length, strict=True, facet_filter=flt, result_sort=sorts)
                    sarr = (_SortC * len(sorts))()
                    for i, so in enumerate(sorts):
                        sarr[i].facet_offset, sarr[i].facet_type, sarr[i].descending = so[0], N.FACET_TYPES[so[1]], 1 if so[2] else 0
                        if so[1] == "point":
                            sarr[i].base[0], sarr[i].base[1] = so[3]
                    farr, nf = S.Shard.facet_filters(flt) if flt else (None, 0)
                    t = np.ascontiguousarray(terms, np.uint32)
                    doc = np.zeros(length + 1, np.uint64); sc = np.zeros(length + 1, np.float32); meta = np.zeros(4, np.uint64)
                    n = H.ssh_index_search_sorted(ix, P(t, u32p), len(t), qt, off_, length, nf, None if farr is None else C.cast(farr, C.c_void_p),
                                                  C.cast(sarr, C.c_void_p), len(sorts), length + 1, P(doc, u64p), P(sc, f32p), P(meta, u64p))
                    assert int(meta[3]) == 0 and int(meta[1]) == ro.result_count_total
                    assert n == ro.result_count and list(doc[:n]) == [r.doc_id for r in ro.results], (terms, sorts, off_, length)
        # Index::search with NOT terms (lexical mode) == the Python mirror's
        H.ssh_search_lexical_ex.argtypes = [C.c_void_p, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u32p, C.c_uint32,
                                            C.POINTER(C.c_uint16), C.c_uint32, C.c_uint32, u64p, f32p, u64p]
        for terms, neg, qt in (([0, 1], [2], 1), ([0, 1], [2], 0), ([2], [0], 1)):
            ro = pidx.search(terms, None, S.QueryType(qt), S.SearchMode.Lexical, 2, 15, strict=True, not_terms=neg)
            t = np.ascontiguousarray(terms, np.uint32); nt = np.ascontiguousarray(neg, np.uint32); ff = np.zeros(1, np.uint16)
            doc = np.zeros(16, np.uint64); sc = np.zeros(16, np.float32); meta = np.zeros(4, np.uint64)
            n = H.ssh_search_lexical_ex(ix, P(t, u32p), len(t), qt, 2, 15, 2, P(nt, u32p), len(nt), ff.ctypes.data_as(C.POINTER(C.c_uint16)), 0, 16,
                                        P(doc, u64p), P(sc, f32p), P(meta, u64p))
            assert int(meta[3]) == 0 and int(meta[1]) == ro.result_count_total and n == ro.result_count
            assert np.allclose(sc[:n], [r.score for r in ro.results], rtol=1e-6)
    finally:
        H.ssh_index_destroy(ix)
        for sh in pshards:
            sh.close()


def test_cpp_string_facet_rank_column_equals_the_python_mirror(H):
    """string_facet_rank_column (host logic, no GPU): ranks by byte-wise UTF-8 order, equal strings share a rank, the column is
    appended little-endian -- the C++ mirror against the Python mirror, String16 and String32"""
    import seekstorm_amd as S
    from seekstorm_amd import _native as N
    H.ssh_string_rank_column.restype = C.c_uint32
    H.ssh_string_rank_column.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint64, C.c_uint32, C.c_void_p]
    words = ["zeta", "alpha", "Alpha", "beta", "\u00e9clair", "eclair", "omega", "beta", "a", "", "zz", "\u4e2d\u6587", "alpha ", "b", "B", "beta"]
    rng = np.random.default_rng(2)
    for ty, dt, off in (("string16", "<u2", 3), ("string32", "<u4", 1)):
        rec = np.dtype([("pad", "u1", (off,)), ("cat", dt), ("x", "<u2")])
        v = np.zeros(5000, rec)
        v["cat"] = rng.integers(0, len(words), len(v)); v["x"] = rng.integers(0, 1000, len(v))
        raw = np.ascontiguousarray(v.view(np.uint8).reshape(len(v), rec.itemsize))
        want, woff = S.Shard.string_facet_rank_column(raw, off, ty, words)
        blob = b"\0".join(w.encode("utf-8") for w in words) + b"\0"
        out = np.zeros((len(v), rec.itemsize + 4), np.uint8)
        got_off = H.ssh_string_rank_column(raw.ctypes.data, len(v), rec.itemsize, off, N.FACET_TYPES[ty], blob, len(blob), len(words), out.ctypes.data)
        assert got_off == woff == rec.itemsize and np.array_equal(out, want)
