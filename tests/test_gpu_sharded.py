"""The sharded entry points (ss_comm_*, ss_*_search_sharded) on one rank: the exchange is the identity, failures are reported instead of hanging (world-size-2 runs: tests/test_host_cpu.py on gloo)."""
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
    # a deep page (k > SS_MAX_K): the shard's list in passes, the exchange and the merge at that k (tests/test_gpu_deep_pages.py)
    for kd in (1500, 9000):  # (one rank x 9000 > 8192: the rank merge)
        doc, score, cnt, tot = sh.search_lexical_batch(q, kd, reference_shortcuts=False)
        md, ms, mc, mt = comm.search_lexical_sharded(sh, q, kd)
        assert np.array_equal(mc, cnt) and np.array_equal(mt, tot) and int(cnt.max()) > 1024
        for i in range(len(q)):
            assert np.array_equal(md[i, :cnt[i]], doc[i, :cnt[i]].astype(np.uint64)) and np.array_equal(ms[i, :cnt[i]], score[i, :cnt[i]])
            assert np.all(md[i, cnt[i]:] == np.uint64(0xFFFFFFFFFFFFFFFF))
    comm.close()


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
    kd = 2500  # a deep vector page through the exchange
    doc_d, score_d, cnt_d, tot_d = sh.search_vector_batch(qs, kd)
    md, ms, mc, mt = comm.search_vector_sharded(sh, qs, kd)
    assert np.array_equal(mc, cnt_d) and np.array_equal(mt, tot_d) and int(cnt_d.min()) > 1024
    for i in range(nq):
        assert np.array_equal(md[i, :cnt_d[i]], doc_d[i, :cnt_d[i]].astype(np.uint64)) and np.array_equal(ms[i, :cnt_d[i]], score_d[i, :cnt_d[i]])
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
    # a deep hybrid page (k = offset + length > SS_MAX_K): both shard tasks in passes, ONE all-gather, the fusion on the host (ss_merge_results)
    hd, hs, hsrc, hc, ht = comm.search_hybrid_sharded(sh, q, qs, 1500, 40)
    for i in range(nq):
        ro = ix.search(tl[i], qs[i], S.QueryType.Union, S.SearchMode.Hybrid, 1500, 40, normalize_query=False)
        assert hc[i] == len(ro.results) == 40 and hd[i, :hc[i]].tolist() == [r.doc_id for r in ro.results]
        assert np.array_equal(hs[i, :hc[i]], np.array([r.score for r in ro.results], np.float32))
        assert hsrc[i, :hc[i]].tolist() == [int(r.source) for r in ro.results] and int(ht[i]) == ro.result_count_total
    n, us = comm.profile_read()
    assert n == 2 + 1 + 2 + 1 and 0.0 < us < 5e4  # one all-gather per sharded call
    # more queries than one pass over the matrix takes (SS_VEC_BATCH = 64): several passes, still ONE all-gather per call
    nq2 = 150
    qs2 = O.vec_gen(O.VECQ_SEED, 100, nq2, dim)
    doc2, score2, cnt2, tot2 = sh.search_vector_batch(qs2, k)
    md, ms, mc, mt = comm.search_vector_sharded(sh, qs2, k)
    assert np.array_equal(mc, cnt2) and np.array_equal(mt, tot2)
    for i in range(nq2):
        assert np.array_equal(md[i, :cnt2[i]], doc2[i, :cnt2[i]].astype(np.uint64)) and np.array_equal(ms[i, :cnt2[i]], score2[i, :cnt2[i]])
    tl2 = [tl[i % len(tl)] for i in range(nq2)]
    q2 = sh.make_queries(tl2, S.QueryType.Union)
    hd, hs, hsrc, hc, ht = comm.search_hybrid_sharded(sh, q2, qs2, 0, 15)
    for i in (0, 63, 64, 65, 127, 128, nq2 - 1):
        ro = ix.search(tl2[i], qs2[i], S.QueryType.Union, S.SearchMode.Hybrid, 0, 15, normalize_query=False)
        assert hc[i] == len(ro.results) and hd[i, :hc[i]].tolist() == [r.doc_id for r in ro.results]
        assert np.array_equal(hs[i, :hc[i]], np.array([r.score for r in ro.results], np.float32))
    n, us = comm.profile_read()
    assert n == 2  # (the counter was reset by the read above)
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
