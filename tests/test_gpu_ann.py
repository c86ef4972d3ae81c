"""GPU parity tests (-m gpu) of the ANN modes (AnnMode::Nprobe / Similaritythreshold / NprobeSimilaritythreshold,
vector.rs:1300-1392) through the C ABI against the oracle's restatement: the SAME clusters must be selected (medoid
scores are computed in the reference's summation order), then the usual top-k bar applies to the visited records."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-4


@pytest.fixture(scope="module")
def S():
    import seekstorm_amd
    return seekstorm_amd


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def l2(m):
    return np.ascontiguousarray(m / np.linalg.norm(m, axis=1, keepdims=True).astype(np.float32), np.float32)


def clustered(O, seed, level_clusters, dim, lo=40, hi=400, spread=0.35):
    """rows in the reference's file order: level after level, cluster after cluster, centre + noise, L2-normalised;
    cluster sizes are deliberately not multiples of the 128-row tile"""
    rng = np.random.default_rng(seed)
    child, rows = [], []
    for nc in level_clusters:
        for _ in range(nc):
            n = int(rng.integers(lo, hi))
            centre = rng.standard_normal(dim).astype(np.float32)
            centre /= np.linalg.norm(centre)
            pts = centre[None, :] + spread * rng.standard_normal((n, dim)).astype(np.float32) / np.sqrt(dim).astype(np.float32)
            rows.append(pts)
            child.append(n)
    rows = np.concatenate(rows).astype(np.float32)
    return l2(rows), np.asarray(child, np.uint32)


def queries_near(O, rows, seed, nq, noise=0.5):
    rng = np.random.default_rng(seed)
    pick = rng.integers(0, len(rows), nq)
    q = rows[pick] + noise * rng.standard_normal((nq, rows.shape[1])).astype(np.float32) / np.sqrt(rows.shape[1]).astype(np.float32)
    return l2(q.astype(np.float32))


MODES = [("Nprobe", 1, None), ("Nprobe", 3, None), ("Similaritythreshold", 0, 0.50002), ("NprobeSimilaritythreshold", 4, 0.50001)]


def mk_mode(S, name, n, t):
    if name == "Nprobe":
        return S.AnnMode.Nprobe(n)
    if name == "Similaritythreshold":
        return S.AnnMode.Similaritythreshold(t)
    return S.AnnMode.NprobeSimilaritythreshold(n, t)


def oracle_mode_args(S, n, t):
    from seekstorm_amd.search import threshold_raw
    return dict(n_probe=n if n else 0xFFFFFFFF, cluster_threshold_raw=threshold_raw(t))


@pytest.mark.parametrize("mode", MODES, ids=[m[0] + str(m[1]) for m in MODES])
@pytest.mark.parametrize("nq", [64, 5])
def test_ann_i8_parity(S, O, mode, nq):
    """integer dots are exact: the selected clusters, the scores and (outside ties) the ids are the oracle's"""
    lc = [7, 12, 5]
    rows32, child = clustered(O, 11, lc, 128)
    rows = O.quantize_i8(rows32)
    qs = O.quantize_i8(queries_near(O, rows32, 12, nq))
    name, n, t = mode
    # the i8 score is the raw integer dot (~127^2 * cosine): place the cluster threshold inside the medoid scores
    t_raw = None
    am = mk_mode(S, name, n, t)
    kw = oracle_mode_args(S, n, t)
    if t is not None:
        med = rows[np.concatenate([[0], np.cumsum(child.astype(np.int64))[:-1]])].astype(np.int32) @ qs[0].astype(np.int32)
        t_raw = float(np.sort(med)[-6])
        from seekstorm_amd.search import SIMILARITY_NORMALIZATION_64_I8
        tn = (np.float32(t_raw) * SIMILARITY_NORMALIZATION_64_I8 + np.float32(1.0)) / np.float32(2.0)
        am = mk_mode(S, name, n, float(tn))
        kw = oracle_mode_args(S, n, float(tn))
    sh = S.Shard(0)
    sh.upload_vectors_i8(rows)
    with pytest.raises(S.SeekStormHipError):
        sh.search_vector_batch_i8(qs, 10, ann_mode=am)  # no cluster structure declared yet
    sh.set_clusters(lc, child)
    assert sh.cluster_info() == (3, 24)
    k = 20
    doc, score, cnt, tot, ncl = sh.search_vector_batch_i8(qs, k, ann_mode=am, with_clusters=True)
    for i in range(nq):
        od, os_, otot, oobs, oncl = O.vec_search_i8_ann(rows, qs[i], k, lc, child, **kw)
        assert ncl[i] == oncl
        assert cnt[i] == len(od)
        assert np.array_equal(score[i][:cnt[i]], os_)
        if len(od):
            kth = os_[-1]
            assert {int(x) for x, y in zip(doc[i][:cnt[i]], score[i]) if y > kth} == {int(x) for x, y in zip(od, os_) if y > kth}
        assert np.all(doc[i][cnt[i]:] == 0xFFFFFFFF)
        assert tot[i] >= cnt[i]
    sh.close()


@pytest.mark.parametrize("dim,simd", [(128, True), (100, False)])
def test_ann_f32_parity(S, O, dim, simd):
    """f32: medoids are scored in the reference's order (8 fmadd lanes, or sequential for dim % 8 != 0), so the cluster
    choice is the oracle's bit for bit; record scores then meet the usual tolerance"""
    lc = [9, 6, 11, 4]
    rows, child = clustered(O, 21, lc, dim)
    qs = queries_near(O, rows, 22, 40)
    sh = S.Shard(0)
    sh.upload_vectors(rows)
    sh.set_clusters(lc, child)
    k = 25
    for am, kw in [(S.AnnMode.Nprobe(2), dict(n_probe=2)), (S.AnnMode.Nprobe(5), dict(n_probe=5))]:
        doc, score, cnt, tot, ncl = sh.search_vector_batch(qs, k, ann_mode=am, with_clusters=True)
        for i in range(len(qs)):
            od, os_, otot, oobs, oncl = O.vec_search_ann(rows, qs[i], k, lc, child, simd_order=simd, **kw)
            assert ncl[i] == oncl == sum(min(kw["n_probe"], c) for c in lc)
            n = int(cnt[i])
            assert n == len(od)
            assert np.all(score[i][:n - 1] >= score[i][1:n])
            assert np.allclose(score[i][:n], os_, rtol=REL, atol=2e-6)
            band = abs(float(os_[-1])) * REL + 2e-6
            strict = lambda dd, ss: {int(x) for x, y in zip(dd, ss) if y > os_[-1] + band}
            assert strict(doc[i][:n], score[i][:n]) == strict(od, os_)
    sh.close()


def test_ann_every_cluster_equals_all_mode(S, O):
    """n_probe >= clusters of every level visits everything: identical to AnnMode::All, and to a brute-force product"""
    lc = [5, 8]
    rows32, child = clustered(O, 31, lc, 64)
    rows = O.quantize_i8(rows32)
    qs = O.quantize_i8(queries_near(O, rows32, 32, 17))
    sh = S.Shard(0)
    sh.upload_vectors_i8(rows)
    sh.set_clusters(lc, child)
    a = sh.search_vector_batch_i8(qs, 30)
    b = sh.search_vector_batch_i8(qs, 30, ann_mode=S.AnnMode.Nprobe(1000), with_clusters=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert np.all(b[4] == 13)
    full = rows.astype(np.int32) @ qs.astype(np.int32).T
    for i in range(len(qs)):
        assert np.array_equal(b[1][i], np.sort(full[:, i])[::-1][:30].astype(np.float32))
    sh.close()


def test_ann_selection_is_per_query_inside_one_batch(S, O):
    """one pass scans the union of the batch's clusters; a row still counts only for the queries that chose its cluster:
    every query of a batch gets what it gets alone"""
    lc = [16, 16]
    rows32, child = clustered(O, 41, lc, 64, lo=100, hi=300)
    rows = O.quantize_i8(rows32)
    qs = O.quantize_i8(queries_near(O, rows32, 42, 64))
    sh = S.Shard(0)
    sh.upload_vectors_i8(rows)
    sh.set_clusters(lc, child)
    am = S.AnnMode.Nprobe(1)
    doc, score, cnt, tot = sh.search_vector_batch_i8(qs, 10, ann_mode=am)
    starts = np.concatenate([[0], np.cumsum(child.astype(np.int64))])
    cluster_of = np.searchsorted(starts, np.arange(len(rows)), side="right") - 1
    for i in range(64):
        d1, s1, c1, _ = sh.search_vector_batch_i8(qs[i:i + 1], 10, ann_mode=am)
        assert np.array_equal(score[i], s1[0]) and cnt[i] == c1[0]
        got = cluster_of[doc[i][:cnt[i]]]
        assert len(set(got[got < 16])) <= 1 and len(set(got[got >= 16])) <= 1  # one cluster per level
    sh.close()


def test_ann_vector_bin_keeps_clusters_with_dedup_and_tombstones(S, O):
    """a vector.bin as the reference writes it: cluster structure from the file, several records per doc, deleted docs"""
    from oracle import ref_format as RF
    dim = 32
    lc = [4, 3]
    rows, child = clustered(O, 51, lc, dim, lo=30, hi=90)
    rng = np.random.default_rng(52)
    levels, ids, r = [], [], 0
    ci = 0
    for lvl, nc in enumerate(lc):
        clusters = []
        for _ in range(nc):
            recs = []
            for _ in range(int(child[ci])):
                d = int(rng.integers(0, 60))  # doc ids collide inside a level: several records per doc
                recs.append((d, 0, 0, rows[r]))
                ids.append((lvl << 16) | d)
                r += 1
            clusters.append(recs)
            ci += 1
        levels.append(clusters)
    ids = np.asarray(ids, np.uint32)
    sh = S.Shard(0)
    sh.upload_vector_bin(RF.write_vector_bin(levels, dim), dim)
    assert sh.cluster_info() == (2, 7)
    gone = [3, (1 << 16) | 7, 41]
    sh.set_deleted(gone)
    qs = queries_near(O, rows, 53, 9)
    k = 12
    doc, score, cnt, tot, ncl = sh.search_vector_batch(qs, k, ann_mode=S.AnnMode.Nprobe(2), with_clusters=True)
    for i in range(len(qs)):
        od, os_, _, _, oncl = O.vec_search_ann(rows, qs[i], k, lc, child, n_probe=2, row_doc_ids=ids, deleted=gone)
        n = int(cnt[i])
        assert ncl[i] == oncl == 4
        assert n == len(od) and not set(map(int, doc[i][:n])) & set(gone)
        assert len(set(map(int, doc[i][:n]))) == n
        assert np.allclose(score[i][:n], os_, rtol=REL, atol=2e-6)
    sh.close()


def test_ann_many_tiles_large_k(S, O):
    """enough rows for every launch of the chunk schedule, k = 100, a full batch: i8 scores stay bit-exact"""
    lc = [40, 40, 40]
    rows32, child = clustered(O, 61, lc, 64, lo=400, hi=1200)
    rows = O.quantize_i8(rows32)
    qs = O.quantize_i8(queries_near(O, rows32, 62, 64))
    sh = S.Shard(0)
    sh.upload_vectors_i8(rows)
    sh.set_clusters(lc, child)
    k = 100
    doc, score, cnt, tot, ncl = sh.search_vector_batch_i8(qs, k, ann_mode=S.AnnMode.Nprobe(6), with_clusters=True)
    for i in range(0, 64, 7):
        od, os_, _, _, oncl = O.vec_search_i8_ann(rows, qs[i], k, lc, child, n_probe=6)
        assert ncl[i] == oncl == 18 and cnt[i] == len(od)
        assert np.array_equal(score[i][:cnt[i]], os_)
    sh.close()


def test_ann_abi_validation(S, O):
    rows = O.quantize_i8(O.vec_gen(O.VEC_SEED, 0, 1000, 32))
    sh = S.Shard(0)
    sh.upload_vectors_i8(rows)
    with pytest.raises(S.SeekStormHipError):
        sh.set_clusters([2], [400, 500])        # does not cover the rows
    with pytest.raises(S.SeekStormHipError):
        sh.set_clusters([3], [400, 0, 600])     # an empty cluster has no medoid
    with pytest.raises(ValueError):
        sh.set_clusters([3], [400, 600])
    with pytest.raises(ValueError):
        S.AnnMode(0, None)._c()
    sh.set_clusters([1, 1], [400, 600])
    d, s, c, t = sh.search_vector_batch_i8(O.quantize_i8(O.vec_gen(O.VECQ_SEED, 0, 1, 32)), 5, ann_mode=S.AnnMode.Nprobe(1))
    assert c[0] == 5
    sh.upload_vectors_i8(rows[:500])           # a new image drops the old cluster structure
    assert sh.cluster_info() == (0, 0)
    sh.close()


def test_ann_tied_medoids_follow_the_reference_replay(S, O):
    """equal medoid scores at the last TopK slot: which of the tied clusters is visited depends on the order of the pushes
    and on the evicted minimum (vector.rs:410-496) -- the device must visit the oracle's clusters, not just k good ones"""
    lc = [7, 6]
    rows32, child = clustered(O, 71, lc, 64, lo=60, hi=200)
    first = np.concatenate([[0], np.cumsum(child.astype(np.int64))[:-1]])
    # level 0: clusters 0, 2, 3, 5 share one medoid record; level 1: clusters 8, 9 and 11, 12 pairwise
    for a, b in ((0, 2), (0, 3), (0, 5), (8, 9), (11, 12)):
        rows32[first[b]] = rows32[first[a]]
    rows = O.quantize_i8(rows32)
    qs = O.quantize_i8(queries_near(O, rows32, 72, 64, noise=1.5))
    sh = S.Shard(0)
    sh.upload_vectors_i8(rows)
    sh.set_clusters(lc, child)
    med = rows[first].astype(np.int32) @ qs.astype(np.int32).T  # [cluster][query]
    straddles = 0
    k = 15
    for n_probe in (1, 2, 3, 5):
        doc, score, cnt, tot, ncl = sh.search_vector_batch_i8(qs, k, ann_mode=S.AnnMode.Nprobe(n_probe), with_clusters=True)
        for i in range(64):
            m0 = np.sort(med[:7, i])[::-1]
            straddles += int(m0[n_probe - 1] == m0[n_probe])
            od, os_, _, _, oncl = O.vec_search_i8_ann(rows, qs[i], k, lc, child, n_probe=n_probe)
            assert ncl[i] == oncl and cnt[i] == len(od)
            assert np.array_equal(score[i][:cnt[i]], os_)
    assert straddles > 20  # the tie really sits on the boundary for many (query, n_probe) pairs
    sh.close()


@pytest.mark.parametrize("i8", [False, True], ids=["f32", "i8"])
def test_vector_field_filter_all_and_ann(S, O, i8):
    """field_filter of search_vector_shard (vector.rs:1225-1237, 1397-1400): records of other fields are skipped before
    they are scored -- in AnnMode::All and inside the clusters an ANN mode visits (medoids are scored whatever their field)"""
    lc = [8, 7]
    rows32, child = clustered(O, 81, lc, 64, lo=80, hi=260)
    n = len(rows32)
    rng = np.random.default_rng(82)
    rf = rng.integers(0, 4, n).astype(np.uint16)      # four indexed fields, interleaved like field x chunk records
    ids = (np.arange(n) // 2).astype(np.uint32)       # two records per doc: the filter acts before the dedup
    q32 = queries_near(O, rows32, 83, 20)
    rows, qs = (O.quantize_i8(rows32), O.quantize_i8(q32)) if i8 else (rows32, q32)
    sh = S.Shard(0)
    (sh.upload_vectors_i8 if i8 else sh.upload_vectors)(rows, row_doc_ids=ids)
    search = sh.search_vector_batch_i8 if i8 else sh.search_vector_batch
    ora = O.vec_search_i8_ann if i8 else O.vec_search_ann
    with pytest.raises(S.SeekStormHipError):
        search(qs, 10, field_filter=[1])             # no field ids declared
    sh.set_fields(rf)
    sh.set_clusters(lc, child)
    k = 15
    for am, lcc, kw in ((None, (None, None), {}), (S.AnnMode.Nprobe(3), (lc, child), dict(n_probe=3))):
        for fields in ([2], [0, 3], [0, 1, 2, 3]):
            doc, score, cnt, tot, ncl = search(qs, k, ann_mode=am, field_filter=fields, with_clusters=True)
            for i in range(len(qs)):
                od, os_, otot, oobs, oncl = ora(rows, qs[i], k, lcc[0], lcc[1], row_doc_ids=ids, row_field=rf, fields=fields, **kw)
                c = int(cnt[i])
                assert c == len(od) and ncl[i] == oncl
                if i8:
                    assert np.array_equal(score[i][:c], os_)
                else:
                    assert np.allclose(score[i][:c], os_, rtol=REL, atol=2e-6)
                assert len(set(map(int, doc[i][:c]))) == c
                if fields != [0, 1, 2, 3]:  # every returned doc has a record of a listed field
                    assert all(any(rf[r] in fields for r in (2 * int(d), min(2 * int(d) + 1, n - 1))) for d in doc[i][:c])
    # every field listed == no filter
    a = search(qs, k)
    b = search(qs, k, field_filter=[0, 1, 2, 3])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    sh.close()


def test_vector_bin_field_ids_feed_the_filter(S, O):
    from oracle import ref_format as RF
    dim = 32
    rows, child = clustered(O, 91, [3], dim, lo=40, hi=80)
    rng = np.random.default_rng(92)
    recs, rf, r = [], [], 0
    clusters = []
    for c in child:
        cl = []
        for _ in range(int(c)):
            f = int(rng.integers(0, 3))
            cl.append((r, f, 0, rows[r]))
            rf.append(f)
            r += 1
        clusters.append(cl)
    sh = S.Shard(0)
    sh.upload_vector_bin(RF.write_vector_bin([clusters], dim), dim)
    qs = queries_near(O, rows, 93, 6)
    doc, score, cnt, tot = sh.search_vector_batch(qs, 10, field_filter=[1])
    for i in range(len(qs)):
        od, os_, _, oobs, _ = O.vec_search_ann(rows, qs[i], 10, None, None, row_field=np.asarray(rf, np.uint16), fields=[1])
        assert cnt[i] == len(od) and np.allclose(score[i][:cnt[i]], os_, rtol=REL, atol=2e-6)
        assert all(rf[int(d)] == 1 for d in doc[i][:cnt[i]])
    sh.close()


def test_ann_many_clusters_per_level_take_the_slow_selection(S, O):
    """more clusters in a level than the LDS selection holds (4096), and an n_probe beyond its TopK array (1024): the
    thread-per-level replay serves them -- same clusters as the oracle"""
    rng = np.random.default_rng(101)
    dim, C = 32, 5000
    child = rng.integers(1, 4, C).astype(np.uint32)
    rows32 = l2(rng.standard_normal((int(child.sum()), dim)).astype(np.float32))
    rows = O.quantize_i8(rows32)
    qs = O.quantize_i8(queries_near(O, rows32, 102, 6))
    sh = S.Shard(0)
    sh.upload_vectors_i8(rows)
    sh.set_clusters([C], child)
    for n_probe in (10, 2000):
        doc, score, cnt, tot, ncl = sh.search_vector_batch_i8(qs, 20, ann_mode=S.AnnMode.Nprobe(n_probe), with_clusters=True)
        for i in range(len(qs)):
            od, os_, _, _, oncl = O.vec_search_i8_ann(rows, qs[i], 20, [C], child, n_probe=n_probe)
            assert ncl[i] == oncl == n_probe and cnt[i] == len(od) and np.array_equal(score[i][:cnt[i]], os_)
    sh.close()


@pytest.mark.parametrize("i8", [False, True])
def test_observed_vector_count_in_every_mode(S, O, i8):
    """observed_vector_count = TopK::push calls (vector.rs:421, 1510): the records of the VISITED clusters whose field the filter
    lists (1397-1400) and whose doc is not tombstoned (1450-1452) -- counted on the device (SS_ANN_REPORT_OBSERVED: three words
    per query) in AnnMode::All, with a field filter, in the ANN modes, with tombstones, over more than one 64-query batch"""
    lc = [8, 7, 5]
    rows32, child = clustered(O, 181, lc, 64, lo=70, hi=240)
    n = len(rows32)
    rng = np.random.default_rng(182)
    rf = rng.integers(0, 3, n).astype(np.uint16)
    ids = (np.arange(n) // 2).astype(np.uint32)       # two records per doc
    q32 = queries_near(O, rows32, 183, 70)            # two device batches
    rows, qs = (O.quantize_i8(rows32), O.quantize_i8(q32)) if i8 else (rows32, q32)
    sh = S.Shard(0)
    (sh.upload_vectors_i8 if i8 else sh.upload_vectors)(rows, row_doc_ids=ids)
    sh.set_fields(rf)
    sh.set_clusters(lc, child)
    search = sh.search_vector_batch_i8 if i8 else sh.search_vector_batch
    ora = O.vec_search_i8_ann if i8 else O.vec_search_ann
    gone = [int(x) for x in rng.choice(n // 2, 40, replace=False)]
    k = 10
    for deleted in ([], gone):
        sh.set_deleted(deleted)
        for am, lcc, kw in ((None, (None, None), {}), (S.AnnMode.Nprobe(2), (lc, child), dict(n_probe=2)),
                            (S.AnnMode.Nprobe(6), (lc, child), dict(n_probe=6))):
            for fields in ((), [1], [0, 2]):
                doc, score, cnt, tot, ncl, obs = search(qs, k, ann_mode=am, field_filter=fields, with_observed=True)
                plain = search(qs, k, ann_mode=am, field_filter=fields, with_clusters=True)
                assert np.array_equal(plain[0], doc) and np.array_equal(plain[1], score) and np.array_equal(plain[4], ncl)  # the report changes nothing else
                for i in range(0, len(qs), 3):
                    od, os_, otot, oobs, oncl = ora(rows, qs[i], k, lcc[0], lcc[1], row_doc_ids=ids, row_field=rf if fields else None,
                                                    fields=fields, deleted=deleted, **kw)
                    assert int(obs[i]) == oobs, (i8, am, fields, bool(deleted), i, int(obs[i]), oobs)
                    assert int(ncl[i]) == oncl
                    assert int(cnt[i]) == len(od)
    # the reference's per-shard seam reports it in the ResultObject
    sh.set_deleted(gone)
    ro = sh.search_vector_shard(q32[0], 10, ann_mode=S.AnnMode.Nprobe(2), field_filter=[1])
    oobs = ora(rows, qs[0], 10, lc, child, row_doc_ids=ids, row_field=rf, fields=[1], deleted=gone, n_probe=2)[3]
    assert ro.observed_vector_count == oobs and ro.observed_cluster_count == sum(min(2, c) for c in lc)
    ro = sh.search_vector_shard(q32[0], 10)  # AnnMode::All with tombstones: the live records
    assert ro.observed_vector_count == ora(rows, qs[0], 10, None, None, row_doc_ids=ids, deleted=gone)[3] < n
    sh.set_deleted([])
    assert sh.search_vector_shard(q32[0], 10).observed_vector_count == n
    sh.close()


@pytest.mark.parametrize("dim", [256, 768])
def test_ann_sparse_batches_match_oracle(S, O, dim):
    """batches of more than 32 queries under Nprobe on an f32 image whose dim is a multiple of 256 take the sparse kernel (per tile only
    the queries that selected one of its clusters are multiplied, on the VALU): same clusters, same answers as the oracle, with a
    field filter, several records per doc and tombstones; a second chunk of 8 queries per tile (every query on the same clusters)"""
    lc = [40, 36]
    rows, child = clustered(O, 301 + dim, lc, dim, lo=30, hi=150)
    n = len(rows)
    rng = np.random.default_rng(302)
    rf = rng.integers(0, 3, n).astype(np.uint16)
    ids = (np.arange(n) // 2).astype(np.uint32)
    qs = queries_near(O, rows, 303, 48)
    sh = S.Shard(0)
    sh.upload_vectors(rows, row_doc_ids=ids)
    sh.set_fields(rf)
    sh.set_clusters(lc, child)
    gone = [int(x) for x in rng.choice(n // 2, 30, replace=False)]
    k = 20
    for deleted in ([], gone):
        sh.set_deleted(deleted)
        for npb in (2, 3):
            for fields in ((), [1]):
                doc, score, cnt, tot, ncl, obs = sh.search_vector_batch(qs, k, ann_mode=S.AnnMode.Nprobe(npb), field_filter=fields, with_observed=True)
                for i in range(len(qs)):
                    od, os_, otot, oobs, oncl = O.vec_search_ann(rows, qs[i], k, lc, child, row_doc_ids=ids, row_field=rf if fields else None,
                                                                 fields=fields, deleted=deleted, n_probe=npb, simd_order=True)
                    c = int(cnt[i])
                    assert c == len(od) and int(ncl[i]) == oncl and int(obs[i]) == oobs, (dim, npb, fields, i)
                    assert np.allclose(score[i][:c], os_, rtol=REL, atol=2e-6)
                    band = abs(float(os_[-1])) * REL + 2e-6
                    strict = lambda dd, ss: {int(x) for x, y in zip(dd, ss) if y > os_[-1] + band}
                    assert strict(doc[i][:c], score[i][:c]) == strict(od, os_)
    sh.set_deleted([])
    # 40 copies of ONE query: every tile of its clusters concerns 40 queries -- five chunks of 8 per tile
    same = np.repeat(qs[:1], 40, axis=0)
    doc, score, cnt, tot = sh.search_vector_batch(same, k, ann_mode=S.AnnMode.Nprobe(2))
    od, os_, _, _, _ = O.vec_search_ann(rows, qs[0], k, lc, child, row_doc_ids=ids, n_probe=2)
    for i in range(40):
        assert int(cnt[i]) == len(od) and np.array_equal(doc[i], doc[0]) and np.array_equal(score[i], score[0])
    assert np.allclose(score[0][:len(od)], os_, rtol=REL, atol=2e-6)
    sh.close()


def test_ann_searches_on_two_streams_do_not_share_their_selection(S, O):
    """device-resident ANN searches queued on two streams without a sync in between: the per-shard selection state (cluster
    bitmaps, tile list) is handed from one to the other by an event wait inside the library, so each gets its own answer"""
    import ctypes as C
    import torch
    from seekstorm_amd import _native as N
    lc = [12, 10, 9]
    rows, child = clustered(O, 401, lc, 128, lo=200, hi=900)
    qa, qb = queries_near(O, rows, 402, 24), queries_near(O, rows, 403, 24)
    sh = S.Shard(0)
    sh.upload_vectors(rows)
    sh.set_clusters(lc, child)
    k = 15
    dev = torch.device("cuda", 0)
    L = S.lib()
    mode = S.AnnMode.Nprobe(2)._c()

    def bufs():
        return (torch.empty((24, k), dtype=torch.int32, device=dev), torch.empty((24, k), dtype=torch.float32, device=dev),
                torch.empty((24,), dtype=torch.int32, device=dev), torch.empty((24,), dtype=torch.int64, device=dev),
                torch.empty((24,), dtype=torch.int32, device=dev))
    ta, tb = torch.from_numpy(qa).to(dev), torch.from_numpy(qb).to(dev)
    ref = {}
    for name, t in (("a", ta), ("b", tb)):  # one at a time
        o = bufs()
        N.check(L.ss_vec_search_ann_dev(sh._h, 24, t.data_ptr(), k, N.FLT_MIN_NEG, C.addressof(mode), o[0].data_ptr(), o[1].data_ptr(),
                                        o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr(), None), "ann")
        N.check(L.ss_shard_sync(sh._h), "sync")
        ref[name] = [x.cpu().numpy().copy() for x in o]
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    for rnd in range(6):
        oa, ob = bufs(), bufs()
        for t, o, st in ((ta, oa, sa), (tb, ob, sb), (ta, oa, sa), (tb, ob, sb)):  # interleaved, nothing waits on the host
            N.check(L.ss_vec_search_ann_dev(sh._h, 24, t.data_ptr(), k, N.FLT_MIN_NEG, C.addressof(mode), o[0].data_ptr(), o[1].data_ptr(),
                                            o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr(), C.c_void_p(st.cuda_stream)), "ann")
        torch.cuda.synchronize()
        for name, o in (("a", oa), ("b", ob)):
            for x, y in zip(ref[name], o):
                assert np.array_equal(x, y.cpu().numpy()), (rnd, name)
    sh.close()
