"""GPU tests (-m gpu) added in round 2: device-side validation of ops_mask, Count with null list outputs, the RCCL exchange
behind the C ABI, the all_terms_frequent shortcut under a facet filter."""
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


def test_all_terms_frequent_is_off_under_a_facet_filter(S, O):
    """add_result.rs:2096-2100: the shortcut applies only when !facet_filtered -- with a facet filter every match is scored,
    so the answer is the exact top-k of the filtered match set, not the tf >= 10 subset"""
    rng = np.random.default_rng(78)
    n_docs = 60_000
    dl = O.lex_doclen(n_docs)
    lists = []
    for df in (40_000, 33_000):
        d = np.sort(rng.choice(n_docs, df, replace=False)).astype(np.uint32)
        lists.append((d, np.minimum(rng.geometric(0.25, df), 700).astype(np.uint16)))
    offs = np.zeros(3, np.uint64)
    offs[1:] = np.cumsum([len(l[0]) for l in lists])
    docs, tfs = np.concatenate([l[0] for l in lists]), np.concatenate([l[1] for l in lists])
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, offs, docs, tfs)
    osh = O.Shard(n_docs, dl, offs, docs, tfs)
    facet = rng.integers(0, 256, n_docs).astype(np.uint8)
    sh.upload_facets(facet.reshape(n_docs, 1))
    keep = (facet >= 16) & (facet < 240)
    q = sh.make_queries([[0, 1]], S.QueryType.Intersection)
    assert sh.mark_all_terms_frequent(q, 10)["op"][0] >> 31  # the condition itself holds
    # unfiltered: the shortcut changes the answer
    d0, s0, c0, t0 = sh.search_lexical_batch(q, 10)
    sc_d, sc_s, sc_t = osh.search_exhaustive([0, 1], O.OP_AND, 10, reference_shortcuts=True)
    ex_d, ex_s, _ = osh.search_exhaustive([0, 1], O.OP_AND, 10)
    assert np.allclose(s0[0][:c0[0]], sc_s, rtol=1e-4) and not np.array_equal(sc_d, ex_d)
    # filtered: exact top-k over the docs that pass
    osh.set_deleted(np.nonzero(~keep)[0])
    fd, fs, ft = osh.search_exhaustive([0, 1], O.OP_AND, 10)
    for strat in (0, 1):
        sh.set_strategy(strat)
        d1, s1, c1, t1 = sh.search_lexical_batch(q, 10, facet_filter=[(0, "u8", 16, 240)])
        assert int(t1[0]) == ft and c1[0] == len(fd)
        assert np.allclose(s1[0][:c1[0]], fs, rtol=1e-4) and set(d1[0][:c1[0]].tolist()) == set(fd.tolist())
    sh.close()


def test_comm_allgather_merge_single_rank(S, O, lex):
    """ss_comm_create / ss_topk_allgather_merge (RCCL behind the C ABI) with a group of one: pack + all-gather + merge must
    equal ss_topk_merge_dev of the same lists (global id = local * 1 + 0)"""
    import torch
    from seekstorm_amd import _native as N
    from seekstorm_amd import distributed as D
    sh, osh, n_docs = lex
    dev = torch.device("cuda", 0)
    comm = D.ShardComm(0, 1, 0)
    r, n, d = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    N.check(N.lib().ss_comm_info(comm._h, C.byref(r), C.byref(n), C.byref(d)), "ss_comm_info")
    assert (r.value, n.value, d.value) == (0, 1, 0)
    q = sh.make_queries([[3, 7, 11], [5, 9], [4]], S.QueryType.Union)
    k = 10
    doc, score, cnt, tot = sh.search_lexical_batch(q, k)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev).view(dt)
    td, ts, tc = t(doc.view(np.int32), torch.int32), t(score, torch.float32), t(cnt.view(np.int32), torch.int32)
    st = torch.cuda.current_stream(dev)
    for _ in range(2):  # second call reuses the communicator's buffers
        md, ms, mc = comm.allgather_merge(td, ts, tc, k, st.cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(mc.cpu().numpy().view(np.uint32), cnt)
        for i in range(len(q)):
            assert np.array_equal(md[i, :cnt[i]].cpu().numpy(), doc[i, :cnt[i]].astype(np.int64))
            assert np.array_equal(ms[i, :cnt[i]].cpu().numpy(), score[i, :cnt[i]])
    rd, rs, rc = D.merge_gathered_device(td[None], ts[None], tc[None], st.cuda_stream, 0)
    torch.cuda.synchronize()
    assert torch.equal(rd, md) and torch.equal(rs, ms) and torch.equal(rc, mc)
    comm.close()


def test_search_sharded_single_rank_equals_plain_search(S, O, lex):
    """ss_bm25_search_sharded with a communicator of one shard: search + all-gather + all-reduce + merge = ss_bm25_search with
    u64 ids; Count carries the totals only"""
    from seekstorm_amd import distributed as D
    sh, osh, n_docs = lex
    comm = D.ShardComm(0, 1, 0)
    q = sh.make_queries([[3, 7, 11], [5, 9], [4], [2, 6]], S.QueryType.Union)
    q2 = sh.make_queries([[3, 7], [5, 9], [1, 4], [2, 6]], S.QueryType.Intersection)
    for qq in (q, q2):
        doc, score, cnt, tot = sh.search_lexical_batch(qq, 10)
        for _ in range(2):
            md, ms, mc, mt = comm.search_lexical_sharded(sh, qq, 10)
            assert np.array_equal(mc, cnt) and np.array_equal(mt, tot)
            for i in range(len(qq)):
                assert np.array_equal(md[i, :cnt[i]], doc[i, :cnt[i]].astype(np.uint64))
                assert np.array_equal(ms[i, :cnt[i]], score[i, :cnt[i]])
        _, _, _, ct = comm.search_lexical_sharded(sh, qq, 0, result_type=int(S.ResultType.Count))
        assert np.array_equal(ct, tot)
    comm.close()


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
