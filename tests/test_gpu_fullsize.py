"""Full-size parity (-m gpu): the BASELINE.json configurations at their real sizes against the oracle.

C2: 10 M docs, 3-term unions, top-10 -- 32 of the bench's 1000 queries: doc ids outside the tie band, scores 1e-4 relative,
    exact result_count_total; strategies AUTO (pruned) and EXHAUSTIVE; the oracle is the reference-structured dispatch
    (union_docid_3) on the same shard, regenerated on the host.
C3: 10 M x 768 cosine top-100 -- 4 queries against a STREAMED oracle scan (rows regenerated slice by slice, running TopK),
    f32 and i8.
C4: BM25 top-100 + cosine top-100 + RRF against the oracle's lists and so_merge.
C5 shape: shards of ONE generator stream (doc g -> shard g % S) merged on the device against the unsharded corpus.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_FULL = 10_000_000


@pytest.fixture(scope="module")
def S():
    import seekstorm_amd
    return seekstorm_amd


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def F():
    from oracle import fullsize
    return fullsize


@pytest.fixture(scope="module")
def c2(S, O):
    import bench
    tl, th = bench.make_c2_queries(O, 1000)
    sh = S.Shard(0)
    sh.synth_lexical(O.LEX_SEED, N_FULL, th, O.len_table())
    yield sh, tl, th
    sh.close()


def test_c2_full_size_against_oracle(S, O, F, c2):
    from seekstorm_amd import _native as N
    sh, tl, th = c2
    ns, k = 32, 10
    q = sh.make_queries(tl[:ns], S.QueryType.Union)
    ans, osh, remap = F.c2_answers(N_FULL, tl[:ns], th, k, O.OP_OR, O.RT_TOPKCOUNT)
    assert abs(osh.avgdl - sh.lexical_info()["avgdl"]) < 1e-3
    for strat in (N.BM25_AUTO, N.BM25_EXHAUSTIVE):
        sh.set_strategy(strat)
        for rt in (S.ResultType.Topk, S.ResultType.TopkCount):
            doc, score, cnt, tot = sh.search_lexical_batch(q, k, rt)
            for i in range(ns):
                od, os_, otot = ans[i]
                F.check_topk(doc[i, :cnt[i]], score[i, :cnt[i]], od, os_, 1e-4, f"C2 strategy {strat} rt {int(rt)} query {i}")
                if rt == S.ResultType.TopkCount:
                    assert int(tot[i]) == otot
    sh.set_strategy(N.BM25_AUTO)
    # the union_scan formulation (the reference's > 10-term path) answers the same: the two oracle restatements agree
    for i in range(0, ns, 8):
        od, os_, otot = osh.search([remap[t] for t in tl[i]], O.OP_OR, k, O.RT_TOPKCOUNT)
        assert otot == ans[i][2] and np.allclose(os_, ans[i][1], rtol=1e-6)


def test_c2_full_size_intersections_and_counts(S, O, F, c2):
    """2- and 3-term intersections at 10 M docs: exact counts and bit-exact doc-id sets of the top-k (ids outside ties)"""
    sh, tl, th = c2
    ns, k = 8, 10
    pairs = [t[1:] for t in tl[:ns]] + [t for t in tl[:ns]]
    q = sh.make_queries(pairs, S.QueryType.Intersection)
    doc, score, cnt, tot = sh.search_lexical_batch(q, k, S.ResultType.TopkCount)
    ans, _, _ = F.c2_answers(N_FULL, pairs, th, k, O.OP_AND, O.RT_TOPKCOUNT)
    for i in range(len(pairs)):
        od, os_, otot = ans[i]
        assert int(tot[i]) == otot, (i, int(tot[i]), otot)
        F.check_topk(doc[i, :cnt[i]], score[i, :cnt[i]], od, os_, 1e-4, f"C2 AND query {i}")


@pytest.fixture(scope="module")
def c3(S, O):
    dim = 768
    sh = S.Shard(0)
    qs = O.vec_gen(O.VECQ_SEED, 0, 64, dim)
    yield sh, qs, dim
    sh.close()


def test_c3_full_size_streamed_oracle_f32(S, O, F, c3):
    sh, qs, dim = c3
    k, nsv = 100, 4
    sh.synth_vectors(O.VEC_SEED, N_FULL, dim)
    doc, score, cnt, tot = sh.search_vector_batch(qs, k)
    ref = F.c3_answers(N_FULL, dim, qs[:nsv], k, slice_rows=32768)
    for i in range(nsv):
        assert cnt[i] == k
        F.check_topk(doc[i], score[i], ref[i][0], ref[i][1], 1e-4, f"C3 f32 query {i}")


def test_c3_full_size_streamed_oracle_i8(S, O, F, c3):
    sh, qs, dim = c3
    k, nsv = 100, 4
    q8 = O.quantize_i8(qs)
    sh.synth_vectors_i8(O.VEC_SEED, N_FULL, dim)
    doc, score, cnt, tot = sh.search_vector_batch_i8(q8, k)
    ref = F.c3_answers(N_FULL, dim, qs[:nsv], k, slice_rows=32768, i8=True)
    for i in range(nsv):
        assert cnt[i] == k and np.array_equal(score[i], ref[i][1])  # integer dot products: ==
        F.check_topk(doc[i], score[i], ref[i][0], ref[i][1], 0.0, f"C3 i8 query {i}")


def test_c4_full_size_hybrid(S, O, F, c2, c3):
    """hybrid at full size through the planner's pieces: BM25 top-100 (10 M docs) + cosine top-100 (10 M x 768) + RRF"""
    import torch
    sh, tl, th = c2
    vsh, qs, dim = c3
    kh, ns = 100, 4
    vsh.synth_vectors(O.VEC_SEED, N_FULL, dim)
    q = sh.make_queries(tl[:ns], S.QueryType.Union)
    ld, ls, lc, _ = sh.search_lexical_batch(q, kh, S.ResultType.Topk)
    vd, vs, vc, _ = vsh.search_vector_batch(qs[:ns], kh)
    lans, _, _ = F.c2_answers(N_FULL, tl[:ns], th, kh, O.OP_OR, O.RT_TOPK)
    vans = F.c3_answers(N_FULL, dim, qs[:ns], kh, slice_rows=32768)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev).view(dt)
    od, os_, osrc, ocnt = S.rrf_merge_device(t(ld.view(np.int32), torch.int32), t(lc.view(np.int32), torch.int32), t(vd.view(np.int32), torch.int32),
                                             t(vc.view(np.int32), torch.int32), 0, kh, st.cuda_stream)
    torch.cuda.synchronize()
    for i in range(ns):
        F.check_topk(ld[i, :lc[i]], ls[i, :lc[i]], lans[i][0], lans[i][1], 1e-4, f"C4 lexical query {i}")
        F.check_topk(vd[i, :vc[i]], vs[i, :vc[i]], vans[i][0], vans[i][1], 1e-4, f"C4 vector query {i}")
        md, ms, _ = O.merge(2, (ld[i, :lc[i]].astype(np.uint64), ls[i, :lc[i]]), (vd[i, :vc[i]].astype(np.uint64), vs[i, :vc[i]]), 0, kh)
        n = int(ocnt[i])
        assert n == len(md) and np.array_equal(od[i, :n].cpu().numpy().astype(np.uint64), md)
        assert np.allclose(os_[i, :n].cpu().numpy(), ms, rtol=1e-6)
        # and the oracle's own lists fuse to the same ids wherever the two sides agree outside ties
        md2, ms2, _ = O.merge(2, (lans[i][0].astype(np.uint64), lans[i][1]), (vans[i][0].astype(np.uint64), vans[i][1]), 0, kh)
        assert np.allclose(np.sort(ms2)[::-1][:10], np.sort(ms)[::-1][:10], rtol=1e-6)


def test_partitioned_generator_shards_merge_to_the_whole(S, O, F):
    """C5's construction at a size the oracle answers quickly: S = 3 shards of ONE generator stream (doc g -> shard g % S,
    local id g // S, index.rs:5284) -- each shard equals the oracle's shard of the host-regenerated stream, and the device
    merge of the per-shard lists carries the global ids local * S + shard (search.rs:1671)"""
    import bench
    n_shard, Sn, k = 400_000, 3, 10
    tl, th = bench.make_c2_queries(O, 16)
    tab = O.len_table()
    dim = 64
    qs = O.vec_gen(O.VECQ_SEED, 0, 4, dim)
    all_ans = []
    for sid in range(Sn):
        sh = S.Shard(0, shard_id=sid)
        sh.synth_partition(sid, Sn)
        sh.synth_lexical(O.LEX_SEED, n_shard, th, tab)
        q = sh.make_queries(tl, S.QueryType.Union)
        doc, score, cnt, tot = sh.search_lexical_batch(q, k, S.ResultType.TopkCount)
        ans, osh, _ = F.c2_answers(n_shard, tl, th, k, O.OP_OR, O.RT_TOPKCOUNT, part=(sid, Sn))
        for i in range(len(tl)):
            assert int(tot[i]) == ans[i][2]
            F.check_topk(doc[i, :cnt[i]], score[i, :cnt[i]], ans[i][0], ans[i][1], 1e-4, f"shard {sid} query {i}")
        all_ans.append((doc, score, cnt))
        # vectors of the same partition
        sh.synth_vectors(O.VEC_SEED, 5000, dim)
        g = O.vec_gen(O.VEC_SEED, 0, 5000 * Sn, dim)[sid::Sn]
        assert np.allclose(sh.read_rows(0, 5000), g, rtol=0, atol=1e-7)  # same hash stream; <= 1 ulp from sqrt / div rounding
        vd, vs, vc, _ = sh.search_vector_batch(qs, 10)
        for i in range(len(qs)):
            od, os_, _, _ = O.vec_search(g, qs[i], 10)
            F.check_topk(vd[i, :vc[i]], vs[i, :vc[i]], od, os_, 1e-4, f"shard {sid} vector query {i}")
        sh.close()
    # cross-shard merge of the lexical lists == oracle merge of the oracle's shard lists
    for i in range(len(tl)):
        ids, sc = [], []
        for sid in range(Sn):
            d, s_, c = all_ans[sid]
            ids += [int(x) * Sn + sid for x in d[i, :c[i]]]
            sc += [float(x) for x in s_[i, :c[i]]]
        md, ms, _ = S.merge_results(S.SearchMode.Lexical, (ids, sc), None, 0, k)
        od, os_, _ = O.merge(0, (np.array(ids, np.uint64), np.array(sc, np.float32)), None, 0, k)
        assert np.array_equal(np.asarray(md, np.uint64), od) and np.allclose(ms, os_)
