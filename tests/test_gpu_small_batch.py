"""The one-launch path of small host-pointer lexical batches (csrc/bm25_small.hip; the reference's call shape is ONE query per
call, search.rs:1637-1743): its answers must be those of the staged pipeline (expand -> probe -> merge -> copies) BIT FOR BIT and
agree with the CPU oracle -- unions / intersections of 1..4 terms, NOT terms, tombstones, exact counts, k from 1 to 128,
1..64 queries per call -- and batches outside its shape must still be answered (by the staged pipeline).  The threshold seeds
(the K-th largest weight of every list, bm_kth_kernel) are checked through what they must never do: cost a query a result."""
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


N_DOCS = 300_000
DFS = [0.004, 0.012, 0.03, 0.05, 0.09, 0.15, 0.0005, 0.22]


def _corpus(O, seed=11):
    rng = np.random.default_rng(seed)
    lens = np.clip(np.round(np.exp(np.log(120) + 0.6 * rng.standard_normal(N_DOCS))), 8, 2000).astype(np.int64)
    lut = {int(x): int(O.lib().so_int_to_byte4(int(x))) for x in np.unique(lens)}
    doclen = np.array([lut[int(x)] for x in lens], np.uint8)
    offs, docs, tfs = [0], [], []
    for df in DFS:
        n = max(1, int(round(df * N_DOCS)))
        docs.append(np.sort(rng.choice(N_DOCS, n, replace=False)).astype(np.uint32))
        tfs.append(rng.geometric(0.55, n).clip(1, 300).astype(np.uint16))
        offs.append(offs[-1] + n)
    return doclen, np.asarray(offs, np.uint64), np.concatenate(docs), np.concatenate(tfs)


@pytest.fixture(scope="module")
def world(S, O):
    dl, offs, docs, tfs = _corpus(O)
    sh = S.Shard(0)
    sh.upload_lexical(N_DOCS, dl, offs, docs, tfs)
    osh = O.Shard(N_DOCS, dl, offs, docs, tfs)
    yield sh, osh
    sh.close()


def _dev_search(S, sh, q, k, rt, ops):
    """the staged pipeline, by construction: device-resident queries through ss_bm25_search_dev"""
    import ctypes as C
    import torch
    from seekstorm_amd import _native as N
    dev = torch.device("cuda", 0)
    nq = len(q)
    qd = torch.from_numpy(q.view(np.uint8).reshape(nq, -1).copy()).to(dev)
    kk = max(k, 1)
    o_doc = torch.empty((nq, kk), dtype=torch.int32, device=dev); o_score = torch.empty((nq, kk), dtype=torch.float32, device=dev)
    o_cnt = torch.empty((nq,), dtype=torch.int32, device=dev); o_tot = torch.empty((nq,), dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev)
    N.check(N.lib().ss_bm25_search_dev(sh._h, nq, qd.data_ptr(), k, int(rt), ops, o_doc.data_ptr(), o_score.data_ptr(), o_cnt.data_ptr(),
                                       o_tot.data_ptr(), C.c_void_p(st.cuda_stream)), "ss_bm25_search_dev")
    torch.cuda.synchronize()
    return (o_doc.cpu().numpy().view(np.uint32), o_score.cpu().numpy(), o_cnt.cpu().numpy().view(np.uint32), o_tot.cpu().numpy().view(np.uint64))


def _queries(rng, n, n_terms, n_not=0):
    out, nots = [], []
    for _ in range(n):
        t = rng.choice(len(DFS), n_terms + n_not, replace=False)
        out.append([int(x) for x in t[:n_terms]])
        nots.append([int(x) for x in t[n_terms:]])
    return out, nots


@pytest.mark.parametrize("shape", ["or", "and", "or_not", "and_not", "tombstones"])
def test_one_launch_equals_staged_pipeline_and_oracle(S, O, world, shape):
    sh, osh = world
    rng = np.random.default_rng({"or": 1, "and": 2, "or_not": 3, "and_not": 4, "tombstones": 5}[shape])
    gone = np.unique(rng.integers(0, N_DOCS, N_DOCS // 50)).astype(np.uint64) if shape == "tombstones" else np.zeros(0, np.uint64)
    sh.set_deleted(gone); osh.set_deleted([int(x) for x in gone])
    try:
        for nq, nt, k in ((1, 1, 10), (1, 3, 10), (7, 2, 1), (64, 3, 10), (33, 4, 32), (5, 3, 33), (16, 2, 100), (3, 4, 128), (65, 3, 10), (200, 2, 10), (256, 3, 32)):
            n_not = 2 if shape.endswith("_not") else 0
            if nt + n_not > len(DFS):
                continue
            lists, nots = _queries(rng, nq, nt, n_not)
            is_and = shape.startswith("and")
            qt = S.QueryType.Intersection if is_and else S.QueryType.Union
            q = sh.make_queries(lists, qt, nots if n_not else None)
            ops = (1 if is_and else 2) | ((nt + n_not) << 8) | (nt << 16) | (n_not << 24)
            for rt in (S.ResultType.Topk, S.ResultType.TopkCount):
                before = sh.one_launch_batches()
                got = sh.search_lexical_batch(q, k, rt, reference_shortcuts=False)
                assert sh.one_launch_batches() == before + 1, "the batch did not take the one-launch path"
                ref = _dev_search(S, sh, q, k, rt, ops)
                assert np.array_equal(got[2], ref[2]), (shape, nq, nt, k, rt)
                for i in range(nq):
                    c = int(got[2][i])
                    assert np.array_equal(got[0][i][:c], ref[0][i][:c]) and np.array_equal(got[1][i][:c], ref[1][i][:c]), (shape, nq, nt, k, rt, i)
                if rt == S.ResultType.TopkCount:
                    assert np.array_equal(got[3], ref[3]), (shape, nq, nt, k)
                # the oracle on a few queries of the batch: exact counts, scores within 1e-4, ids outside the k-th score's tie band
                for i in range(min(nq, 3)):
                    od, os_, otot = osh.search_exhaustive(lists[i], O.OP_AND if (is_and and nt > 1) else O.OP_OR, k, not_terms=nots[i] if n_not else ())
                    c = int(got[2][i])
                    assert c == len(od), (shape, i, c, len(od))
                    assert np.allclose(got[1][i][:c], os_, rtol=1e-4)
                    if rt == S.ResultType.TopkCount:
                        assert int(got[3][i]) == otot
                    if c:
                        band = abs(float(os_[-1])) * 2e-4
                        clear = lambda dd, ss: {int(x) for x, y in zip(dd, ss) if y > os_[-1] + band}
                        assert clear(got[0][i][:c], got[1][i][:c]) <= set(od.tolist()) and clear(od, os_) <= set(got[0][i][:c].tolist())
    finally:
        sh.set_deleted([]); osh.set_deleted([])


def test_batches_outside_the_shape_take_the_staged_pipeline(S, O, world):
    sh, osh = world
    rng = np.random.default_rng(3)
    # 257 queries (up to 256: launches of 64 back to back), 5 scored terms, k = 129, a Count request: all answered, none by the one-launch path
    for lists, k, rt in ((_queries(rng, 257, 3)[0], 10, S.ResultType.Topk), (_queries(rng, 4, 5)[0], 10, S.ResultType.Topk),
                         (_queries(rng, 4, 3)[0], 129, S.ResultType.Topk), (_queries(rng, 4, 3)[0], 10, S.ResultType.Count)):
        q = sh.make_queries(lists, S.QueryType.Union)
        before = sh.one_launch_batches()
        d, s, c, t = sh.search_lexical_batch(q, k, rt, reference_shortcuts=False)
        assert sh.one_launch_batches() == before
        for i in range(min(len(lists), 2)):
            od, os_, otot = osh.search_exhaustive(lists[i], O.OP_OR, k)
            if rt == S.ResultType.Count:
                assert int(t[i]) == otot
            else:
                assert int(c[i]) == len(od) and np.allclose(s[i][:c[i]], os_, rtol=1e-4)
    # a call split by shape: 200 queries, every fifth one of 5 terms (staged), the others by launches of 64 -- answers in the callers' order
    lists = _queries(rng, 200, 3)[0]
    for j in range(0, 200, 5):
        lists[j] = _queries(rng, 1, 5)[0][0]
    q = sh.make_queries(lists, S.QueryType.Union)
    before = sh.one_launch_batches()
    d, s, c, t = sh.search_lexical_batch(q, 10, S.ResultType.TopkCount, reference_shortcuts=False)
    assert sh.one_launch_batches() == before + 1
    for i in list(range(0, 12)) + [63, 64, 65, 128, 198, 199]:
        od, os_, otot = osh.search_exhaustive(lists[i], O.OP_OR, 10)
        assert int(t[i]) == otot and int(c[i]) == len(od) and np.allclose(s[i][:c[i]], os_, rtol=1e-4), i
    # an invalid query keeps its error code on this path too (term id out of range)
    from seekstorm_amd import _native as N
    q = sh.make_queries([[0, 1]], S.QueryType.Union)
    q["term"][0, 0] = 10_000
    with pytest.raises(N.SeekStormHipError):
        sh.search_lexical_batch(q, 10, S.ResultType.Topk, reference_shortcuts=False)
    # and the shard still answers afterwards (the per-query state of the one-launch path is left clean)
    q = sh.make_queries([[0, 1, 2]], S.QueryType.Union)
    d, s, c, t = sh.search_lexical_batch(q, 10, S.ResultType.TopkCount, reference_shortcuts=False)
    od, os_, otot = osh.search_exhaustive([0, 1, 2], O.OP_OR, 10)
    assert int(t[0]) == otot and np.allclose(s[0][:c[0]], os_, rtol=1e-4)


def test_threshold_seeds_never_cost_a_result(S, O):
    """lists whose K largest weights sit in a handful of docs, ties AT the seed, lists shorter than K, k on both sides of 10 / 100:
    seeded searches (default) return what the exhaustive strategy returns"""
    rng = np.random.default_rng(77)
    n_docs = 120_000
    lens = rng.choice([20, 21, 300], n_docs)  # few distinct lengths: large groups of equal weights, i.e. ties at the K-th weight
    lut = {int(x): int(O.lib().so_int_to_byte4(int(x))) for x in np.unique(lens)}
    dl = np.array([lut[int(x)] for x in lens], np.uint8)
    offs, docs, tfs = [0], [], []
    for n, big in ((9, 0), (60, 3), (5000, 12), (40000, 200), (15000, 1000), (100, 100)):
        d = np.sort(rng.choice(n_docs, n, replace=False)).astype(np.uint32)
        tf = np.ones(n, np.uint16)
        if big:
            tf[rng.choice(n, min(big, n), replace=False)] = 25
        docs.append(d); tfs.append(tf); offs.append(offs[-1] + n)
    offs = np.asarray(offs, np.uint64); docs = np.concatenate(docs); tfs = np.concatenate(tfs)
    from seekstorm_amd import _native as N
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, offs, docs, tfs)
    osh = O.Shard(n_docs, dl, offs, docs, tfs)
    try:
        lists = [[0], [1], [2], [3], [0, 1], [2, 3], [1, 2, 3], [0, 2, 4], [3, 4, 5], [2, 3, 4, 5], [5], [4, 5]]
        for k in (1, 9, 10, 11, 99, 100, 101, 128):
            q = sh.make_queries(lists, S.QueryType.Union)
            seeded = sh.search_lexical_batch(q, k, S.ResultType.TopkCount, reference_shortcuts=False)       # one launch, host seeds
            big = sh.make_queries(lists * 22, S.QueryType.Union)                                          # 264 queries: staged, device seeds
            staged = sh.search_lexical_batch(big, k, S.ResultType.TopkCount, reference_shortcuts=False)
            sh.set_strategy(N.BM25_EXHAUSTIVE)
            exh = sh.search_lexical_batch(q, k, S.ResultType.TopkCount, reference_shortcuts=False)
            sh.set_strategy(N.BM25_AUTO)
            for i, tl in enumerate(lists):
                c = int(exh[2][i])
                assert int(seeded[2][i]) == c == int(staged[2][i]), (k, tl, int(seeded[2][i]), int(staged[2][i]), c)
                assert np.array_equal(seeded[1][i][:c], exh[1][i][:c]) and np.array_equal(staged[1][i][:c], exh[1][i][:c]), (k, tl)
                assert int(seeded[3][i]) == int(exh[3][i]) == int(staged[3][i])
                od, os_, otot = osh.search_exhaustive(tl, O.OP_OR, k)
                assert c == len(od) and np.allclose(seeded[1][i][:c], os_, rtol=1e-4) and int(seeded[3][i]) == otot
    finally:
        sh.close()


def test_one_launch_on_an_image_with_several_indexed_fields(S, O):
    """several indexed fields (BM25F): a query without a field filter reads the merged per-term lists, i.e. one list per term --
    the one-launch path serves it; a query WITH a field filter does not take it.  Answers == the staged pipeline's, and the oracle's"""
    rng = np.random.default_rng(9)
    n_docs, n_fields = 90_000, 3
    lens = np.clip(np.round(np.exp(np.log([12, 200, 8])[:, None] + 0.5 * rng.standard_normal((n_fields, n_docs)))), 1, 60000).astype(np.int64)
    lut = {int(x): int(O.lib().so_int_to_byte4(int(x))) for x in np.unique(lens)}
    dl = np.vectorize(lut.get)(lens).astype(np.uint8)
    boost = [2.0, 1.0, 0.5]
    offs, D, F, T = [0], [], [], []
    for df in (0.01, 0.03, 0.06, 0.12, 0.004):
        d = np.sort(rng.choice(n_docs, int(df * n_docs), replace=False)).astype(np.uint32)
        m = rng.random((len(d), n_fields)) < np.array([0.3, 0.9, 0.2])
        m[~m.any(1), 1] = True
        di, fi = np.nonzero(m)
        D.append(d[di]); F.append(fi.astype(np.uint8)); T.append(np.minimum(rng.geometric(0.5, len(di)), 300).astype(np.uint16))
        offs.append(offs[-1] + len(di))
    offs = np.asarray(offs, np.uint64); D = np.concatenate(D); F = np.concatenate(F); T = np.concatenate(T)
    sh = S.Shard(0)
    try:
        sh.upload_lexical_fields(n_docs, dl, boost, offs, D, F, T)
        if not sh.fields_info()[1]:
            pytest.skip("the image was built without merged lists")
        for qt, oop, is_and in ((S.QueryType.Union, O.OP_OR, False), (S.QueryType.Intersection, O.OP_AND, True)):
            lists = [[0, 1], [1, 2, 3], [4], [0, 2, 3, 4], [2, 3]]
            nots = [[], [4], [], [], [0]]
            q = sh.make_queries(lists, qt, nots)
            for rt in (S.ResultType.Topk, S.ResultType.TopkCount):
                before = sh.one_launch_batches()
                got = sh.search_lexical_batch(q, 10, rt, reference_shortcuts=False)
                assert sh.one_launch_batches() == before + 1, "a multi-field batch without a field filter did not take the one-launch path"
                ops = (1 if is_and else 0) | 2 | (5 << 8) | (4 << 16) | (1 << 24)
                ref = _dev_search(S, sh, q, 10, rt, ops)
                assert np.array_equal(got[2], ref[2])
                for i in range(len(lists)):
                    c = int(got[2][i])
                    assert np.array_equal(got[1][i][:c], ref[1][i][:c]) and np.array_equal(got[0][i][:c], ref[0][i][:c]), (qt, rt, i)
                    od, os_, otot, _ = O.search_fields_exhaustive(n_docs, dl, boost, offs, D, F, T, lists[i], oop if len(lists[i]) > 1 else O.OP_OR, 10, nots[i])
                    assert c == len(od) and np.allclose(got[1][i][:c], os_, rtol=1e-4), (qt, rt, i)
                    if rt == S.ResultType.TopkCount:
                        assert int(got[3][i]) == otot and np.array_equal(got[3], ref[3])
        # under a field filter the query reads (term, field) lists: the staged pipeline
        qf = sh.make_queries([[0, 1]], S.QueryType.Intersection, field_filter=[1])
        before = sh.one_launch_batches()
        sh.search_lexical_batch(qf, 10, S.ResultType.TopkCount, reference_shortcuts=False)
        assert sh.one_launch_batches() == before
    finally:
        sh.close()


# ------------------------------------------------------------------------------------------------ the tiered image (round 6)
def _staged_tiered(S, sh, q, k, rt):
    """the staged tiered pipeline by construction: a device-resident batch that declares sparse-tier terms (ops_mask bit 28) is split on
    the host and never takes the one-launch path"""
    return _dev_search(S, sh, q, k, rt, 1 << 28)


def _same_answers(got, ref, rt, S, what):
    assert np.array_equal(got[2], ref[2]), (what, "counts", got[2], ref[2])
    for i in range(len(got[2])):
        c = int(got[2][i])
        assert np.array_equal(got[0][i][:c], ref[0][i][:c]) and np.array_equal(got[1][i][:c], ref[1][i][:c]), (what, i, got[0][i][:c], ref[0][i][:c])
    if rt == S.ResultType.TopkCount:
        assert np.array_equal(got[3], ref[3]), (what, "totals", got[3], ref[3])


def test_one_launch_on_a_tiered_image(S, O):
    """a realistic vocabulary: a few dense lists, rare terms in the SPARSE tier.  Small batches naming sparse terms take ONE launch (role 3
    of bm25_small_kernel) -- unions and intersections of 1 .. 4 terms of either tier, NOT terms, tombstones, exact counts, k <= 32, batches
    mixing tiered and all-dense queries: bit-identical to the staged tiered pipeline, and within tolerance of the oracle"""
    rng = np.random.default_rng(61)
    n_docs = 200_000
    dens = [0.2, 0.09, 0.05, 0.02, 0.011, 0.004]
    rare = [3000, 1400, 800, 300, 120, 60, 25, 9, 3, 1]
    lens = np.clip(np.round(np.exp(np.log(120) + 0.6 * rng.standard_normal(n_docs))), 8, 2000).astype(np.int64)
    lut = {int(x): int(O.lib().so_int_to_byte4(int(x))) for x in np.unique(lens)}
    dl = np.array([lut[int(x)] for x in lens], np.uint8)
    offs, docs, tfs = [0], [], []
    for n in [int(x * n_docs) for x in dens] + rare:
        docs.append(np.sort(rng.choice(n_docs, n, replace=False)).astype(np.uint32))
        tfs.append(rng.geometric(0.5, n).clip(1, 200).astype(np.uint16))
        offs.append(offs[-1] + n)
    offs, docs, tfs = np.asarray(offs, np.uint64), np.concatenate(docs), np.concatenate(tfs)
    nd, nt_all = len(dens), len(dens) + len(rare)
    e = int(offs[nd])
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, offs[:nd + 1], docs[:e], tfs[:e])
    assert sh.append_sparse(offs[nd:] - offs[nd], docs[e:], tfs[e:]) == nd
    osh = O.Shard(n_docs, dl, offs, docs, tfs)
    D, R = list(range(nd)), list(range(nd, nt_all))

    def batch(n, n_terms, n_sparse, n_not_dense=0, n_not_sparse=0):
        lists, nots = [], []
        for _ in range(n):
            sp = [int(x) for x in rng.choice(R, n_sparse + n_not_sparse, replace=False)]
            de = [int(x) for x in rng.choice(D, n_terms - n_sparse + n_not_dense, replace=False)]
            t = sp[:n_sparse] + de[:n_terms - n_sparse]
            rng.shuffle(t)
            lists.append([int(x) for x in t])
            nots.append(de[n_terms - n_sparse:] + sp[n_sparse:])
        return lists, nots

    gone = np.unique(rng.integers(0, n_docs, n_docs // 40)).astype(np.uint64)
    try:
        for deleted in (False, True):
            sh.set_deleted(gone if deleted else [])
            osh.set_deleted([int(x) for x in gone] if deleted else [])
            for qt, oop in ((S.QueryType.Union, O.OP_OR), (S.QueryType.Intersection, O.OP_AND)):
                shapes = [(1, 1, 1, 0, 0), (1, 3, 1, 0, 0), (7, 2, 1, 0, 0), (64, 3, 1, 0, 0), (33, 4, 2, 0, 0), (5, 4, 4, 0, 0), (16, 3, 2, 2, 0), (9, 2, 2, 1, 0)]
                if qt == S.QueryType.Intersection:
                    shapes += [(6, 3, 1, 1, 1), (4, 2, 2, 0, 2)]  # a sparse NOT list where the sparse role meets it
                for nq, nt, ns, nnd, nns in shapes:
                    lists, nots = batch(nq, nt, ns, nnd, nns)
                    if nq >= 7:  # mixed: some all-dense queries ride along
                        for j in range(0, nq, 3):
                            lists[j] = [int(x) for x in rng.choice(D, min(nt, 4), replace=False)]
                            nots[j] = [t for t in nots[j] if t < nd and t not in lists[j]]
                    q = sh.make_queries(lists, qt, nots)
                    for k in (1, 10, 32):
                        for rt in (S.ResultType.Topk, S.ResultType.TopkCount):
                            before = sh.one_launch_batches()
                            got = sh.search_lexical_batch(q, k, rt, reference_shortcuts=False)
                            assert sh.one_launch_batches() == before + 1, ("not one launch", qt, nq, nt, ns, nnd, nns, k, rt)
                            ref = _staged_tiered(S, sh, q, k, rt)
                            _same_answers(got, ref, rt, S, (qt, nq, nt, ns, nnd, nns, k, rt, deleted))
                            for i in range(min(nq, 4)):
                                od, os_, otot = osh.search_exhaustive(lists[i], oop if len(lists[i]) > 1 else O.OP_OR, k, not_terms=nots[i])
                                c = int(got[2][i])
                                assert c == len(od) and np.allclose(got[1][i][:c], os_, rtol=1e-4), (qt, lists[i], nots[i], k)
                                if rt == S.ResultType.TopkCount:
                                    assert int(got[3][i]) == otot, (qt, lists[i], nots[i], int(got[3][i]), otot)
                                if c < k:
                                    assert set(got[0][i][:c].tolist()) == set(od.tolist())
        # outside the shape: a union with a SPARSE NOT term, k = 33 -- answered by the staged pipeline
        sh.set_deleted([]); osh.set_deleted([])
        for lists, nots, k in (([[0, 1]], [[nd + 2]], 10), ([[0, nd]], [[]], 33)):
            before = sh.one_launch_batches()
            d, s_, c, t = sh.search_lexical_batch(sh.make_queries(lists, S.QueryType.Union, nots), k, reference_shortcuts=False)
            assert sh.one_launch_batches() == before
            od, os_, otot = osh.search_exhaustive(lists[0], O.OP_OR, k, not_terms=nots[0])
            assert int(t[0]) == otot and int(c[0]) == len(od) and np.allclose(s_[0][:c[0]], os_, rtol=1e-4)
    finally:
        sh.close()


def test_one_launch_phrases_naming_sparse_terms(S, O):
    """phrases of <= 4 unique terms that name a sparse-tier term -- the common case of a quoted name -- are role 3 of the one launch: bit-identical
    to the staged sparse phrase kernel, and the oracle's answers; NOT terms of either tier, tombstones, counts, mixed with set queries"""
    from test_gpu_phrase import _corpus
    n_docs = 120_000
    dfs = [30_000, 22_000, 9_000, 500, 300, 1_500, 40]
    nd = 3
    plant = [([0, 3], 80), ([3, 4], 40), ([1, 3, 2], 60), ([5, 0, 5], 50), ([4, 4], 30), ([0, 1], 300), ([6, 5, 3, 0], 12), ([2, 5], 70), ([3, 0, 1, 3], 25)]
    dl, offs, docs, tfs, positions = _corpus(O, n_docs, dfs, 23, plant)
    e = int(offs[nd]); pe = int(tfs[:e].astype(np.int64).sum())
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, offs[:nd + 1], docs[:e], tfs[:e], positions[:pe])
    assert sh.append_sparse(offs[nd:] - offs[nd], docs[e:], tfs[e:], positions=positions[pe:]) == nd
    osh = O.Shard(n_docs, dl, offs, docs, tfs)
    osh.set_positions(positions)
    cases = [([0, 3], []), ([3, 0], []), ([3, 4], []), ([1, 3, 2], []), ([5, 0, 5], []), ([4, 4], []), ([6, 5, 3, 0], []), ([2, 5], []), ([3, 0, 1, 3], []),
             ([0, 3], [1]), ([2, 5], [3, 0]), ([3, 4], [5]), ([5, 2], [])]
    gone = list(range(3, n_docs, 61))
    try:
        for deleted in (False, True):
            sh.set_deleted(gone if deleted else [])
            osh.set_deleted(gone if deleted else [])
            for sel in (list(range(len(cases))), [0], [6, 8]):
                q = sh.make_queries([cases[i][0] for i in sel], S.QueryType.Phrase, [cases[i][1] for i in sel])
                for k in (10, 32):
                    for rt in (S.ResultType.TopkCount, S.ResultType.Topk):
                        before = sh.one_launch_batches()
                        got = sh.search_lexical_batch(q, k, rt, reference_shortcuts=False)
                        assert sh.one_launch_batches() == before + 1, ("phrases: not one launch", sel, k, rt)
                        ref = _dev_search(S, sh, q, k, rt, (1 << 28) | 16)
                        _same_answers(got, ref, rt, S, ("phrase", sel, k, rt, deleted))
                        for j, ci in enumerate(sel):
                            ph, neg = cases[ci]
                            uniq = list(dict.fromkeys(ph))
                            od, os_, otot = osh.search_phrase(uniq, [uniq.index(w) for w in ph], n_docs)
                            drop = set()
                            for t in neg:
                                drop |= set(docs[int(offs[t]):int(offs[t + 1])].tolist())
                            keep = [x for x, d in enumerate(od.tolist()) if d not in drop]
                            c = int(got[2][j])
                            assert c == min(k, len(keep)) and np.allclose(got[1][j][:c], os_[keep][:k], rtol=1e-4), (ph, neg, k)
                            if rt == S.ResultType.TopkCount:
                                assert int(got[3][j]) == len(keep), (ph, neg, int(got[3][j]), len(keep))
        # a batch mixing sparse phrases with set queries of both tiers: still one launch
        sh.set_deleted([]); osh.set_deleted([])
        qm = sh.make_queries([[0, 3], [0, 3], [3, 4], [0, 1], [3, 0]], [S.QueryType.Phrase, S.QueryType.Union, S.QueryType.Intersection, S.QueryType.Union, S.QueryType.Phrase])
        before = sh.one_launch_batches()
        tm = sh.search_lexical_batch(qm, 10, reference_shortcuts=False)[3]
        assert sh.one_launch_batches() == before + 1
        assert int(tm[0]) == osh.search_phrase([0, 3], [0, 1], 10)[2] and int(tm[1]) == osh.search_exhaustive([0, 3], O.OP_OR, 10)[2]
        assert int(tm[2]) == osh.search_exhaustive([3, 4], O.OP_AND, 10)[2] and int(tm[3]) == osh.search_exhaustive([0, 1], O.OP_OR, 10)[2]
        assert int(tm[4]) == osh.search_phrase([3, 0], [0, 1], 10)[2]
    finally:
        sh.close()


def test_one_launch_on_a_tiered_image_with_several_indexed_fields(S, O):
    """merged lists + a sparse tier of merged weights (BM25F): unions, intersections and phrases naming sparse terms in one launch"""
    from test_gpu_parity import _fields_corpus, _check_topk
    from test_gpu_phrase import _corpus_fields
    n_docs, n_fields, boost = 60_000, 3, [2.0, 1.0, 0.5]
    dfs = [18_000, 9_000, 5_000, 2_500, 400, 90, 30, 7]
    nd = 4
    plant = [([0, 4], 0, 30), ([4, 5], 2, 20), ([1, 4, 2], 1, 25), ([6, 0], 0, 10)]
    dl, offs, docs, fields, tfs, positions = _corpus_fields(O, n_docs, n_fields, dfs, 31, plant, [(0, 1, 20)])
    e = int(offs[nd]); pe = int(tfs[:e].astype(np.int64).sum())
    sh = S.Shard(0)
    sh.upload_lexical_fields(n_docs, dl, boost, offs[:nd + 1], docs[:e], fields[:e], tfs[:e], positions[:pe])
    assert sh.append_sparse_fields(offs[nd:] - offs[nd], docs[e:], fields[e:], tfs[e:], positions=positions[pe:]) == nd
    try:
        cases = [([0, 4], []), ([6, 1, 2], []), ([5, 6], []), ([4], []), ([7, 5, 0], []), ([3, 5], [1]), ([4, 5, 0, 1], [2])]
        for qt, oop in ((S.QueryType.Union, O.OP_OR), (S.QueryType.Intersection, O.OP_AND)):
            q = sh.make_queries([c[0] for c in cases], qt, [c[1] for c in cases])
            for k in (10, 32):
                before = sh.one_launch_batches()
                got = sh.search_lexical_batch(q, k, S.ResultType.TopkCount, reference_shortcuts=False)
                assert sh.one_launch_batches() == before + 1
                ref = _staged_tiered(S, sh, q, k, S.ResultType.TopkCount)
                _same_answers(got, ref, S.ResultType.TopkCount, S, ("3f", qt, k))
                for i, (pos, neg) in enumerate(cases):
                    od, os_, otot, _ = O.search_fields_exhaustive(n_docs, dl, boost, offs, docs, fields, tfs, pos, oop if len(pos) > 1 else O.OP_OR, k, neg, ())
                    assert int(got[3][i]) == otot, (qt, pos, neg, int(got[3][i]), otot)
                    _check_topk(got[0][i], got[1][i], got[2][i], od, os_)
        phrases = [[0, 4], [4, 5], [1, 4, 2], [6, 0]]
        q = sh.make_queries(phrases, S.QueryType.Phrase)
        before = sh.one_launch_batches()
        got = sh.search_lexical_batch(q, 10, S.ResultType.TopkCount, reference_shortcuts=False)
        assert sh.one_launch_batches() == before + 1
        ref = _dev_search(S, sh, q, 10, S.ResultType.TopkCount, (1 << 28) | 16)
        _same_answers(got, ref, S.ResultType.TopkCount, S, "3f phrases")
        for i, ph in enumerate(phrases):
            od, os_, otot = O.search_fields_phrase(n_docs, dl, boost, offs, docs, fields, tfs, positions, ph, list(range(len(ph))), 10)
            assert int(got[3][i]) == otot >= 10 and np.allclose(got[1][i][:got[2][i]], os_, rtol=1e-4), (ph, int(got[3][i]), otot)
    finally:
        sh.close()
