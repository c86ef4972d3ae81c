"""Unions of 11..32 terms: the reference sends them to union_blockid -> union_scan_32 (search.rs:3497-3520,
union.rs:598-805: a 32-bit match mask per doc, score = sum over the matched terms, exact union count).  Through the C ABI: Topk /
TopkCount / Count, NOT lists, tombstones, k of 10 and 100, against the oracle's restatement of that dispatch (so_search_lex_ref ->
search_or) and against the plain definition (search_exhaustive).  More than SS_MAX_QUERY_TERMS terms: SS_EINVAL."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N_DOCS = 40_000
N_TERMS = 40


@pytest.fixture(scope="module")
def S():
    import seekstorm_amd
    return seekstorm_amd


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def world(S, O):
    rng = np.random.default_rng(2024)
    lens = np.clip(np.round(np.exp(np.log(120) + 0.6 * rng.standard_normal(N_DOCS))), 8, 2000).astype(np.int64)
    lut = {int(x): int(O.lib().so_int_to_byte4(int(x))) for x in np.unique(lens)}
    dl = np.array([lut[int(x)] for x in lens], np.uint8)
    dfs = np.exp(rng.uniform(np.log(0.0005), np.log(0.12), N_TERMS))
    offs, docs, tfs = [0], [], []
    for df in dfs:
        n = max(2, int(round(df * N_DOCS)))
        docs.append(np.sort(rng.choice(N_DOCS, n, replace=False)).astype(np.uint32))
        tfs.append(rng.geometric(0.55, n).clip(1, 200).astype(np.uint16))
        offs.append(offs[-1] + n)
    offs = np.asarray(offs, np.uint64); docs = np.concatenate(docs); tfs = np.concatenate(tfs)
    sh = S.Shard(0)
    sh.upload_lexical(N_DOCS, dl, offs, docs, tfs)
    osh = O.Shard(N_DOCS, dl, offs, docs, tfs)
    yield sh, osh
    sh.close()


def _check(O, got, i, od, os_, otot, rt, S, what):
    doc, score, cnt, tot = got
    if rt != S.ResultType.Topk:
        assert int(tot[i]) == otot, (what, int(tot[i]), otot)
    if rt == S.ResultType.Count:
        return
    c = int(cnt[i])
    assert c == len(od), (what, c, len(od))
    assert np.allclose(score[i][:c], os_, rtol=1e-4), what
    if c:
        band = abs(float(os_[-1])) * 2e-4
        clear = lambda dd, ss: {int(x) for x, y in zip(dd, ss) if y > os_[-1] + band}
        assert clear(doc[i][:c], score[i][:c]) <= set(od.tolist()) and clear(od, os_) <= set(doc[i][:c].tolist()), what


@pytest.mark.parametrize("deleted", [False, True])
def test_unions_of_11_to_32_terms_match_the_oracle(S, O, world, deleted):
    sh, osh = world
    rng = np.random.default_rng(5 + int(deleted))
    gone = np.unique(rng.integers(0, N_DOCS, N_DOCS // 40)).astype(np.uint64) if deleted else np.zeros(0, np.uint64)
    sh.set_deleted(gone); osh.set_deleted(gone)
    try:
        for nt, n_not in ((11, 0), (12, 2), (16, 0), (24, 1), (32, 0), (29, 3), (10, 0), (7, 0)):
            lists, nots = [], []
            for _ in range(6):
                t = rng.choice(N_TERMS, nt + n_not, replace=False)
                lists.append([int(x) for x in t[:nt]]); nots.append([int(x) for x in t[nt:]])
            q = sh.make_queries(lists, S.QueryType.Union, nots if n_not else None)
            for k in (10, 100):
                for rt in (S.ResultType.Topk, S.ResultType.TopkCount, S.ResultType.Count):
                    got = sh.search_lexical_batch(q, k, rt, reference_shortcuts=False)
                    # the 16-bit many-list scan (bm25_scan16m_kernel, what AUTO runs here) against the f32 tile: the same sums, bit for bit
                    from seekstorm_amd import _native as N
                    sh.set_strategy(N.BM25_EXHAUSTIVE_F32)
                    f32 = sh.search_lexical_batch(q, k, rt, reference_shortcuts=False)
                    sh.set_strategy(N.BM25_AUTO)
                    assert np.array_equal(got[2], f32[2]) and np.array_equal(got[3], f32[3]), (nt, n_not, k, int(rt))
                    for i in range(len(lists)):
                        assert np.array_equal(got[1][i][:got[2][i]], f32[1][i][:got[2][i]]), (nt, n_not, k, int(rt), i)
                    for i in range(len(lists)):
                        what = (nt, n_not, k, int(rt), i, deleted)
                        od, os_, otot = osh.search_exhaustive(lists[i], O.OP_OR, k, not_terms=nots[i])  # the definition
                        _check(O, got, i, od, os_, otot, rt, S, what)
                        if nt > 10 or rt == S.ResultType.Count:  # ... and the reference's own formulation (union_scan's table walk)
                            rd, rs, rtot = osh.search_ref(lists[i], O.OP_OR, k, O.RT_TOPKCOUNT if rt != S.ResultType.Count else O.RT_COUNT, not_terms=nots[i])
                            assert rtot == otot
                            if rt != S.ResultType.Count:
                                _check(O, got, i, rd, rs, rtot, rt, S, what + ("ref",))
    finally:
        sh.set_deleted([]); osh.set_deleted([])


def test_intersections_of_many_terms_and_the_limit(S, O, world):
    sh, osh = world
    from seekstorm_amd import _native as N
    # intersections of 11 / 12 terms over the densest lists: mostly empty, exact counts either way
    order = np.argsort([-osh.df(t) for t in range(N_TERMS)])
    for nt in (11, 12):
        terms = [int(x) for x in order[:nt]]
        q = sh.make_queries([terms], S.QueryType.Intersection)
        got = sh.search_lexical_batch(q, 10, S.ResultType.TopkCount, reference_shortcuts=False)
        od, os_, otot = osh.search_exhaustive(terms, O.OP_AND, 10)
        _check(O, got, 0, od, os_, otot, S.ResultType.TopkCount, S, ("and", nt))
    # 33 terms: the host's own dispatch (the reference ranks, block by block, the 32 lists with the largest block maxima and recounts,
    # union.rs:233-259, 617-624; not modelled) -- reported as SS_ENOTSUP / cpu_dispatch, never as an empty answer or an invalid query
    assert N.SS_MAX_QUERY_TERMS == 32
    with pytest.raises(N.SeekStormHipError) as e:
        sh.make_queries([list(range(33))], S.QueryType.Union)
    assert e.value.code == N.SS_ENOTSUP
    ro = sh.search_lexical_shard(list(range(33)), S.QueryType.Union, 0, 10)
    assert ro.cpu_dispatch and not ro.results
