"""Full-size parity (-m gpu): the BASELINE.json configurations at their real sizes against the oracle.

C2: 10 M docs, 3-term unions, top-10 -- ALL 1000 queries of the bench batch: doc ids outside the tie band, scores 1e-4 relative,
    exact result_count_total; strategies AUTO (pruned) and EXHAUSTIVE, Topk and TopkCount; the oracle is the reference-structured
    dispatch (union_docid_3) on the same shard, regenerated on the host.  256 two-term + 64 three-term intersections likewise.
C3: 10 M x 768 cosine top-100 -- all 64 queries of the batch against a STREAMED oracle scan (rows regenerated slice by slice,
    running TopK), f32 and i8.
C4: BM25 top-100 + cosine top-100 + RRF against the oracle's lists and so_merge, 64 queries.
C5 shape: S = 8 shards of ONE generator stream (doc g -> shard g % 8), 1 M docs + 1 M x 768 rows each, per-shard lists against the
    oracle's shards and the device merge / concat / RRF kernels at k = 10 and 100 against so_merge of the oracle's lists.
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
    """ALL 1000 queries of the bench batch: every strategy x result type against the reference-structured oracle"""
    from seekstorm_amd import _native as N
    sh, tl, th = c2
    ns, k = len(tl), 10
    q = sh.make_queries(tl, S.QueryType.Union)
    ans = F.c2_answers_chunked(N_FULL, tl, th, k, O.OP_OR, O.RT_TOPKCOUNT)
    first = None
    for strat in (N.BM25_AUTO, N.BM25_EXHAUSTIVE):
        sh.set_strategy(strat)
        for rt in (S.ResultType.Topk, S.ResultType.TopkCount):
            doc, score, cnt, tot = sh.search_lexical_batch(q, k, rt)
            if first is None:
                first = (doc.copy(), score.copy())
            assert np.array_equal(score, first[1]) and np.array_equal(doc, first[0])  # the four runs agree bit for bit
            for i in range(ns):
                od, os_, otot = ans[i]
                F.check_topk(doc[i, :cnt[i]], score[i, :cnt[i]], od, os_, 1e-4, f"C2 strategy {strat} rt {int(rt)} query {i}")
                if rt == S.ResultType.TopkCount:
                    assert int(tot[i]) == otot
    sh.set_strategy(N.BM25_AUTO)
    # the union_scan formulation (the reference's > 10-term path) answers the same: the two oracle restatements agree
    ans2, osh, remap = F.c2_answers(N_FULL, tl[:32:8], th, k, O.OP_OR, O.RT_TOPKCOUNT, structured=False)
    assert abs(osh.avgdl - sh.lexical_info()["avgdl"]) < 1e-3
    for j, i in enumerate(range(0, 32, 8)):
        assert ans2[j][2] == ans[i][2] and np.allclose(ans2[j][1], ans[i][1], rtol=1e-6)


def test_c2_full_size_intersections_and_counts(S, O, F, c2):
    """256 two-term and 64 three-term intersections at 10 M docs: exact counts and bit-exact doc-id sets of the top-k (ids
    outside ties), pruned and exhaustive strategies"""
    from seekstorm_amd import _native as N
    import bench
    sh, tl, th = c2
    k = 10
    rng = np.random.default_rng(4321)
    ba, bb = bench.band_terms(th, 0.01, 0.05), bench.band_terms(th, 0.05, 0.20)  # the bench's AND leg: C1's df bands
    pairs = [[int(rng.choice(ba)), int(rng.choice(bb))] for _ in range(256)] + [t for t in tl[:64]]
    q = sh.make_queries(pairs, S.QueryType.Intersection)
    ans = F.c2_answers_chunked(N_FULL, pairs, th, k, O.OP_AND, O.RT_TOPKCOUNT, chunk=160)
    for strat in (N.BM25_AUTO, N.BM25_EXHAUSTIVE):
        sh.set_strategy(strat)
        # the whole (mixed 2- / 3-term) batch, and the two uniform halves on their own: under EXHAUSTIVE a batch whose queries
        # all have the same number of terms takes the 16-bit scan's intersection instance, a mixed one the f32 kernel
        for lo, hi in ((0, len(pairs)), (0, 256), (256, len(pairs))):
            doc, score, cnt, tot = sh.search_lexical_batch(q[lo:hi].copy(), k, S.ResultType.TopkCount)
            for j in range(hi - lo):
                i = lo + j
                od, os_, otot = ans[i]
                assert int(tot[j]) == otot, (strat, i, int(tot[j]), otot)
                F.check_topk(doc[j, :cnt[j]], score[j, :cnt[j]], od, os_, 1e-4, f"C2 AND strategy {strat} query {i}")
                if len(od) < k:  # fewer matches than k: the doc-id SET is the whole intersection -- bit exact
                    assert set(doc[j, :cnt[j]].tolist()) == set(int(x) for x in od)
    sh.set_strategy(N.BM25_AUTO)


@pytest.fixture(scope="module")
def c3(S, O):
    dim = 768
    sh = S.Shard(0)
    qs = O.vec_gen(O.VECQ_SEED, 0, 64, dim)
    yield sh, qs, dim
    sh.close()


@pytest.fixture(scope="module")
def c3_ref(O, F, c3):
    """the streamed oracle's top-100 of ALL 64 queries of the batch over the 10 M x 768 rows (shared by the C3 and C4 tests)"""
    sh, qs, dim = c3
    return F.c3_answers(N_FULL, dim, qs, 100, slice_rows=32768)


def test_c3_full_size_streamed_oracle_f32(S, O, F, c3, c3_ref):
    sh, qs, dim = c3
    k = 100
    sh.synth_vectors(O.VEC_SEED, N_FULL, dim)
    doc, score, cnt, tot = sh.search_vector_batch(qs, k)
    for i in range(len(qs)):
        assert cnt[i] == k
        F.check_topk(doc[i], score[i], c3_ref[i][0], c3_ref[i][1], 1e-4, f"C3 f32 query {i}")


def test_c4_full_size_hybrid(S, O, F, c2, c3, c3_ref):
    """hybrid at full size through the planner's pieces, 64 queries: BM25 top-100 (10 M docs) + cosine top-100 (10 M x 768) + RRF"""
    import torch
    sh, tl, th = c2
    vsh, qs, dim = c3
    kh, ns = 100, 64
    vsh.synth_vectors(O.VEC_SEED, N_FULL, dim)  # (whatever image an earlier test left: < 1 s on the device)
    q = sh.make_queries(tl[:ns], S.QueryType.Union)
    ld, ls, lc, _ = sh.search_lexical_batch(q, kh, S.ResultType.Topk)
    vd, vs, vc, _ = vsh.search_vector_batch(qs[:ns], kh)
    lans, _, _ = F.c2_answers(N_FULL, tl[:ns], th, kh, O.OP_OR, O.RT_TOPK)
    vans = c3_ref
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


def test_c3_full_size_streamed_oracle_i8(S, O, F, c3):
    sh, qs, dim = c3
    k = 100
    q8 = O.quantize_i8(qs)
    sh.synth_vectors_i8(O.VEC_SEED, N_FULL, dim)
    doc, score, cnt, tot = sh.search_vector_batch_i8(q8, k)
    ref = F.c3_answers(N_FULL, dim, qs, k, slice_rows=32768, i8=True)
    for i in range(len(qs)):
        assert cnt[i] == k and np.array_equal(score[i], ref[i][1])  # integer dot products: ==
        F.check_topk(doc[i], score[i], ref[i][0], ref[i][1], 0.0, f"C3 i8 query {i}")


def test_c5_shape_eight_shards_through_the_device_merges(S, O, F):
    """BASELINE configs[4] in shape: S = 8 shards of ONE generator stream (doc / row g -> shard g % 8, local id g // 8,
    index.rs:5284), 1 M docs and 1 M x 768 rows per shard, built one after the other on this GPU.  Every shard's lexical,
    vector lists against the oracle's shard (shard-local N, df, avgdl); then the 8 lists of every query through the device
    merges exactly as an all-gather leaves them -- ss_topk_merge_dev_packed (k = 10 and 100: search.rs:1875-1940, 2098-2119,
    ids local * 8 + shard :1671), ss_topk_concat_dev_packed + ss_rrf_merge_dev (hybrid: RRF over the whole concatenations,
    1962-2035) -- against so_merge of the ORACLE's 8 lists"""
    import torch
    import bench
    from seekstorm_amd import _native as N
    from seekstorm_amd import distributed as D
    Sn, n_shard, dim, nq = 8, 1_000_000, 768, 64
    tl, th = bench.make_c2_queries(O, nq)
    tab = O.len_table()
    qs = O.vec_gen(O.VECQ_SEED, 0, nq, dim)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev)
    T = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev).view(dt)
    got = {10: {"lex": [], "vec": []}, 100: {"lex": [], "vec": []}}
    ref = {10: {"lex": [], "vec": []}, 100: {"lex": [], "vec": []}}
    tot_lex = np.zeros(nq, np.int64)
    ref_tot_lex = np.zeros(nq, np.int64)
    for sid in range(Sn):
        sh = S.Shard(0, shard_id=sid)
        sh.synth_partition(sid, Sn)
        sh.synth_lexical(O.LEX_SEED, n_shard, th, tab)
        sh.synth_vectors(O.VEC_SEED, n_shard, dim)
        q = sh.make_queries(tl, S.QueryType.Union)  # this shard's own idf
        lans = F.c2_answers(n_shard, tl, th, 100, O.OP_OR, O.RT_TOPKCOUNT, part=(sid, Sn))[0]
        vans = F.c3_answers(n_shard, dim, qs, 100, part=(sid, Sn), slice_rows=32768)
        for k in (10, 100):
            ld, ls, lc, lt = sh.search_lexical_batch(q, k, S.ResultType.TopkCount)
            vd, vs, vc, _ = sh.search_vector_batch(qs, k)
            for i in range(nq):
                assert int(lt[i]) == lans[i][2]
                F.check_topk(ld[i, :lc[i]], ls[i, :lc[i]], lans[i][0][:k], lans[i][1][:k], 1e-4, f"shard {sid} k {k} lexical query {i}")
                F.check_topk(vd[i, :vc[i]], vs[i, :vc[i]], vans[i][0][:k], vans[i][1][:k], 1e-4, f"shard {sid} k {k} vector query {i}")
            got[k]["lex"].append((ld, ls, lc)); got[k]["vec"].append((vd, vs, vc))
            ref[k]["lex"].append([(a[0][:k], a[1][:k]) for a in lans]); ref[k]["vec"].append([(a[0][:k], a[1][:k]) for a in vans])
            if k == 10:
                tot_lex += lt.astype(np.int64)
                ref_tot_lex += np.array([a[2] for a in lans], np.int64)
        sh.close()
    assert np.array_equal(tot_lex, ref_tot_lex)  # result_count_total is summed over the shards (search.rs:1884)

    def packed(lists):  # what ONE all-gather of the shards' packed lists leaves on every rank
        return torch.stack([D.pack_topk(T(d.view(np.int32), torch.int32), T(s_, torch.float32), T(c.view(np.int32), torch.int32)) for d, s_, c in lists])

    def cat(lists, i):  # the oracle's per-shard lists of query i appended in shard order with global ids
        ids = np.concatenate([r[i][0].astype(np.uint64) * Sn + sid for sid, r in enumerate(lists)])
        sc = np.concatenate([r[i][1] for r in lists]).astype(np.float32)
        return ids, sc

    for k in (10, 100):
        merged = {}
        for mode, name in ((0, "lex"), (1, "vec")):
            pk = packed(got[k][name])
            md, ms, mc = D.merge_gathered_device_packed(pk, nq, k, st.cuda_stream, 0)
            cd = torch.empty((nq, Sn * k), dtype=torch.int64, device=dev); cs = torch.empty((nq, Sn * k), dtype=torch.float32, device=dev)
            cc = torch.empty((nq,), dtype=torch.int32, device=dev)
            N.check(N.lib().ss_topk_concat_dev_packed(0, nq, Sn, k, pk.data_ptr(), cd.data_ptr(), cs.data_ptr(), cc.data_ptr(), st.cuda_stream), "concat")
            torch.cuda.synchronize()
            merged[name] = (cd, cc)
            md, ms, mc = md.cpu().numpy().astype(np.uint64), ms.cpu().numpy(), mc.cpu().numpy()
            for i in range(nq):
                ids, sc = cat(ref[k][name], i)
                od, os_, _ = O.merge(mode, (ids, sc), (ids, sc), 0, k) if mode == 0 else O.merge(1, None, (ids, sc), 0, k)
                assert mc[i] == len(od)
                F.check_topk(md[i, :mc[i]], ms[i, :mc[i]], od, os_, 1e-4, f"S = 8 merge, {name}, k {k}, query {i}")
                # exactly: the device merge of the DEVICE's lists = so_merge of the same lists (ids, order, scores)
                gids = np.concatenate([d[i, :c[i]].astype(np.uint64) * Sn + sid for sid, (d, s_, c) in enumerate(got[k][name])])
                gsc = np.concatenate([s_[i, :c[i]] for d, s_, c in got[k][name]])
                od2, os2, _ = O.merge(mode, (gids, gsc), (gids, gsc), 0, k) if mode == 0 else O.merge(1, None, (gids, gsc), 0, k)
                assert np.array_equal(md[i, :mc[i]], od2) and np.array_equal(ms[i, :mc[i]], os2)
        # hybrid: RRF over the two whole concatenations, final top-k (offset 0 and an offset inside the list)
        for offset, length in ((0, k), (3, k - 3)):
            hd = torch.empty((nq, length), dtype=torch.int64, device=dev); hs = torch.empty((nq, length), dtype=torch.float32, device=dev)
            hsrc = torch.empty((nq, length), dtype=torch.uint8, device=dev); hc = torch.empty((nq,), dtype=torch.int32, device=dev)
            N.check(N.lib().ss_rrf_merge_dev(0, nq, Sn * k, merged["lex"][0].data_ptr(), merged["lex"][1].data_ptr(), Sn * k, merged["vec"][0].data_ptr(),
                                             merged["vec"][1].data_ptr(), 1, offset, length, hd.data_ptr(), hs.data_ptr(), hsrc.data_ptr(), hc.data_ptr(),
                                             st.cuda_stream), "ss_rrf_merge_dev")
            torch.cuda.synchronize()
            hd_, hs_, hc_ = hd.cpu().numpy().astype(np.uint64), hs.cpu().numpy(), hc.cpu().numpy()
            for i in range(nq):
                gl = (np.concatenate([d[i, :c[i]].astype(np.uint64) * Sn + sid for sid, (d, s_, c) in enumerate(got[k]["lex"])]),
                      np.concatenate([s_[i, :c[i]] for d, s_, c in got[k]["lex"]]))
                gv = (np.concatenate([d[i, :c[i]].astype(np.uint64) * Sn + sid for sid, (d, s_, c) in enumerate(got[k]["vec"])]),
                      np.concatenate([s_[i, :c[i]] for d, s_, c in got[k]["vec"]]))
                od, os_, _ = O.merge(2, gl, gv, offset, length)
                assert hc_[i] == len(od) and np.array_equal(hd_[i, :hc_[i]], od), f"S = 8 hybrid ids, k {k}, query {i}"
                assert np.allclose(hs_[i, :hc_[i]], os_, rtol=1e-6)


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
