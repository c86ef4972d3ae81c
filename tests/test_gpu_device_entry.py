"""The device-pointer entry points and the ABI's argument checks: ops_mask validated on the device, null output lists, searches on several streams of one shard, uploads that check their arrays first."""
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
def lex(S, O):
    n_docs, voc = 300_000, list(range(2500, 4096, 100))
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = O.lex_corpus(n_docs, voc)
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, offs, docs, tfs)
    osh = O.Shard(n_docs, dl, offs, docs, tfs)
    yield sh, osh, n_docs
    sh.close()


def _dev_search(S, sh, q_np, k, rt, ops_mask, null_lists=False):
    import torch
    from seekstorm_amd import _native as N
    dev = torch.device("cuda", 0)
    nq = len(q_np)
    qd = torch.from_numpy(q_np.view(np.uint8).reshape(nq, -1).copy()).to(dev)
    doc = torch.full((nq, max(k, 1)), -1, dtype=torch.int32, device=dev)
    score = torch.zeros((nq, max(k, 1)), dtype=torch.float32, device=dev)
    cnt = torch.full((nq,), 12345, dtype=torch.int32, device=dev)
    tot = torch.full((nq,), -7, dtype=torch.int64, device=dev)
    N.check(N.lib().ss_bm25_search_dev(sh._h, nq, qd.data_ptr(), k, int(rt), ops_mask, None if null_lists else doc.data_ptr(),
                                       None if null_lists else score.data_ptr(), cnt.data_ptr(), tot.data_ptr(), None), "ss_bm25_search_dev")
    N.check(N.lib().ss_shard_sync(sh._h), "sync")
    torch.cuda.synchronize()
    return doc.cpu().numpy().view(np.uint32), score.cpu().numpy(), cnt.cpu().numpy().view(np.uint32), tot.cpu().numpy().view(np.uint64)


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


def test_ops_mask_is_validated_on_the_device(S, O, lex):
    """a device-resident batch is described by ops_mask; a query that contradicts it is flagged count = UINT32_MAX and
    answered as empty instead of running a kernel variant that cannot serve it"""
    sh, osh, n_docs = lex
    q = sh.make_queries([[3, 7, 11], [3, 7], [5, 9, 12, 14]], [S.QueryType.Union, S.QueryType.Intersection, S.QueryType.Union])
    BAD = 0xFFFFFFFF
    # correct description: unions + intersections, up to 4 terms
    doc, score, cnt, tot = _dev_search(S, sh, q, 10, S.ResultType.TopkCount, 1 | 2 | (4 << 8))
    for i, (terms, op) in enumerate((([3, 7, 11], O.OP_OR), ([3, 7], O.OP_AND), ([5, 9, 12, 14], O.OP_OR))):
        od, os_, otot = osh.search_exhaustive(terms, op, 10)
        assert cnt[i] == len(od) and int(tot[i]) == otot and np.allclose(score[i][:cnt[i]], os_, rtol=1e-4)
    # declared union-only: the intersection is refused, the unions are still answered
    doc2, score2, cnt2, tot2 = _dev_search(S, sh, q, 10, S.ResultType.TopkCount, 2 | (4 << 8))
    assert cnt2[1] == BAD and np.all(doc2[1] == BAD) and tot2[1] == 0
    assert np.array_equal(score2[[0, 2]], score[[0, 2]]) and np.array_equal(tot2[[0, 2]], tot[[0, 2]])
    # declared at most 3 terms: the 4-term query is refused
    _, score3, cnt3, _ = _dev_search(S, sh, q, 10, S.ResultType.Topk, 1 | 2 | (3 << 8))
    assert cnt3[2] == BAD and cnt3[0] == 10 and np.array_equal(score3[0], score[0])
    # declared intersection-only: the unions are refused
    _, _, cnt4, _ = _dev_search(S, sh, q, 10, S.ResultType.Topk, 1 | (4 << 8))
    assert cnt4[0] == BAD and cnt4[2] == BAD and cnt4[1] == cnt[1]
    # malformed: a term id outside the vocabulary, no terms
    qb = q.copy()
    qb["term"][0, 1] = 1_000_000
    qb["n_terms"][2] = 0
    _, _, cnt5, _ = _dev_search(S, sh, qb, 10, S.ResultType.Topk, 1 | 2 | (4 << 8))
    assert cnt5[0] == BAD and cnt5[2] == BAD and cnt5[1] == cnt[1]


@pytest.mark.parametrize("strategy", [0, 1])
def test_count_with_null_list_outputs(S, O, lex, strategy):
    """ResultType::Count through the device-pointer entry point with d_out_doc = d_out_score = NULL: counts and totals are
    written whatever the number of partitions (an image with many sub-blocks)"""
    sh, osh, n_docs = lex
    sh.set_strategy(strategy)
    cases = [([3, 7, 11], S.QueryType.Union, O.OP_OR), ([3, 7], S.QueryType.Intersection, O.OP_AND), ([9], S.QueryType.Union, O.OP_OR)]
    q = sh.make_queries([c[0] for c in cases], [c[1] for c in cases])
    _, _, cnt, tot = _dev_search(S, sh, q, 0, S.ResultType.Count, 1 | 2 | (3 << 8), null_lists=True)
    for i, (terms, _, op) in enumerate(cases):
        assert cnt[i] == 0 and int(tot[i]) == osh.search_exhaustive(terms, op, 10)[2]
    sh.set_strategy(0)


def test_bm25_searches_on_two_streams_of_one_shard_overlap_safely(S, O, lex):
    """per-stream workspaces: device-pointer searches queued on two different streams of ONE shard (different batches, different
    batch sizes) run concurrently and each returns what it returns alone"""
    import torch
    from seekstorm_amd import _native as N
    sh, osh, n_docs = lex
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(9)
    batches = []
    for nq in (700, 90):
        tl = [[int(x) for x in rng.choice(16, 3, replace=False)] for _ in range(nq)]
        q = sh.make_queries(tl, S.QueryType.Union)
        batches.append((nq, torch.from_numpy(q.view(np.uint8).reshape(nq, -1).copy()).to(dev), q))
    k = 10
    outs, streams = [], [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    for nq, qd, _ in batches:
        outs.append((torch.empty((nq, k), dtype=torch.int32, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev),
                     torch.empty((nq,), dtype=torch.int32, device=dev), torch.empty((nq,), dtype=torch.int64, device=dev)))
    torch.cuda.synchronize()
    L = N.lib()
    for rep in range(20):  # interleaved launches: both streams busy at the same time
        for (nq, qd, _), (od, os_, oc, ot), st in zip(batches, outs, streams):
            N.check(L.ss_bm25_search_dev(sh._h, nq, qd.data_ptr(), k, N.RT_TOPKCOUNT, 2 | (3 << 8), od.data_ptr(), os_.data_ptr(), oc.data_ptr(),
                                         ot.data_ptr(), C.c_void_p(st.cuda_stream)), "ss_bm25_search_dev")
    torch.cuda.synchronize()
    for (nq, qd, q), (od, os_, oc, ot) in zip(batches, outs):
        d1, s1, c1, t1 = sh.search_lexical_batch(q, k)  # the same batch alone, through the host-pointer entry point
        assert np.array_equal(os_.cpu().numpy(), s1) and np.array_equal(od.cpu().numpy().view(np.uint32), d1)
        assert np.array_equal(ot.cpu().numpy().view(np.uint64), t1)


def test_vector_searches_on_two_streams_of_one_shard_overlap_safely(S, O):
    """per-stream scan buffers: device-pointer vector searches queued on two different streams of ONE shard (different
    batches, interleaved, repeated) each return what they return alone -- f32 and i8"""
    import torch
    from seekstorm_amd import _native as N
    dev = torch.device("cuda", 0)
    n_rows, dim, k = 300_000, 256, 20
    L = S.lib()
    for i8 in (False, True):
        sh = S.Shard(0)
        if i8:
            sh.synth_vectors_i8(O.VEC_SEED, n_rows, dim)
        else:
            sh.synth_vectors(O.VEC_SEED, n_rows, dim)
        qs = [O.vec_gen(O.VECQ_SEED, 0, 64, dim), O.vec_gen(O.VECQ_SEED, 64, 40, dim)]
        if i8:
            qd = [torch.from_numpy(O.quantize_i8(q)).to(dev) for q in qs]
        else:
            qd = [torch.from_numpy(q).to(dev) for q in qs]
        streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        outs = [[torch.empty((len(q), k), dtype=torch.int32, device=dev), torch.empty((len(q), k), dtype=torch.float32, device=dev),
                 torch.empty((len(q),), dtype=torch.int32, device=dev), torch.empty((len(q),), dtype=torch.int64, device=dev)] for q in qs]

        def call(j, st):
            o = outs[j]
            if i8:
                N.check(L.ss_vec_search_i8_dev(sh._h, len(qs[j]), qd[j].data_ptr(), None, k, N.FLT_MIN_NEG, o[0].data_ptr(), o[1].data_ptr(),
                                               o[2].data_ptr(), o[3].data_ptr(), st), "ss_vec_search_i8_dev")
            else:
                N.check(L.ss_vec_search_dev(sh._h, len(qs[j]), qd[j].data_ptr(), k, N.FLT_MIN_NEG, o[0].data_ptr(), o[1].data_ptr(),
                                            o[2].data_ptr(), o[3].data_ptr(), st), "ss_vec_search_dev")
        alone = []
        for j in (0, 1):  # each batch alone on the shard's own stream
            call(j, None)
            N.check(L.ss_shard_sync(sh._h), "sync")
            alone.append([t.clone() for t in outs[j][:3]])
        torch.cuda.synchronize()
        for rep in range(6):
            for t in outs[0] + outs[1]:
                t.zero_()
            torch.cuda.synchronize()
            for j in ((0, 1) if rep % 2 == 0 else (1, 0)):
                call(j, streams[j].cuda_stream)
            torch.cuda.synchronize()
            for j in (0, 1):
                assert torch.equal(outs[j][2], alone[j][2]) and torch.equal(outs[j][0], alone[j][0]) and torch.equal(outs[j][1], alone[j][1]), (i8, rep, j)
        sh.close()


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
