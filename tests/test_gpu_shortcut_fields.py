"""The all_terms_frequent shortcut (intersection.rs:198-209) where it meets facet filters and several indexed fields."""
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


def test_all_terms_frequent_over_several_indexed_fields(S, O):
    """decode_positions_multiterm_multifield's form of the shortcut (add_result.rs:1595-1607, 3111-3122): when N > 256 k and every
    term of an intersection is in at least half of the docs, a doc is counted but ranked only if every term has >= 10 positions in the
    LOWEST field that holds the doc (an embedded pointer -- <= 4 positions -- or a record whose first field has < 10).  Over the
    image's merged lists; off under a field filter (3116); both mirrors mark the query like the reference"""
    rng = np.random.default_rng(91)
    n_docs, n_fields = 50_000, 3
    dl = np.stack([O.lex_doclen(n_docs, seed=O.LEX_SEED + 3 * f) for f in range(n_fields)])
    boost = np.array([1.5, 1.0, 0.5], np.float32)
    dfs = [34_000, 30_000, 27_000, 6_000]
    offs, D, F, T = [0], [], [], []
    for df in dfs:
        for d in np.sort(rng.choice(n_docs, df, replace=False)):
            fs = np.sort(rng.choice(n_fields, size=int(rng.integers(1, n_fields + 1)), replace=False))
            for f in fs:
                D.append(int(d)); F.append(int(f)); T.append(int(min(rng.geometric(0.2), 600)))  # ~13 % of the entries have tf >= 10
        offs.append(len(D))
    offs, D, F, T = np.array(offs, np.uint64), np.array(D, np.uint32), np.array(F, np.uint8), np.array(T, np.uint16)
    sh = S.Shard(0)
    sh.upload_lexical_fields(n_docs, dl, boost, offs, D, F, T)
    assert sh.fields_info() == (3, True, False)
    gone = list(range(1, n_docs, 61))
    sh.set_deleted(gone)
    cases = [[0, 1], [0, 1, 2], [0, 3], [1, 2]]
    flagged = [True, True, False, True]
    q = sh.make_queries(cases, S.QueryType.Intersection)
    rel = 1e-4
    differs = 0
    for k in (10, 150, 250):  # 50 000 > 256 * 150 but not > 256 * 250
        marked = sh.mark_all_terms_frequent(q, k)
        assert [bool(x >> 31) for x in marked["op"]] == [f and k < 250 for f in flagged]
        for strat in (0, 1):
            sh.set_strategy(strat)
            for rt in (S.ResultType.TopkCount, S.ResultType.Topk):
                doc, score, cnt, tot = sh.search_lexical_batch(q, k, rt)
                for i, terms in enumerate(cases):
                    plain = O.search_fields_exhaustive(n_docs, dl, boost, offs, D, F, T, terms, O.OP_AND, k, deleted=gone)
                    if flagged[i] and k < 250:
                        od, os_, otot = O.search_fields_shortcut(n_docs, dl, boost, offs, D, F, T, terms, k, deleted=gone)
                        differs += int(not np.array_equal(od, plain[0]))
                    else:
                        od, os_, otot = plain[:3]
                    if rt == S.ResultType.TopkCount:
                        assert int(tot[i]) == otot == plain[2]
                    n = int(cnt[i])
                    assert n == len(od) and np.allclose(score[i][:n], os_, rtol=rel)
                    if n:
                        band = abs(float(os_[-1])) * rel
                        clear = lambda dd, ss: {int(x) for x, y in zip(dd, ss) if y > os_[-1] + 2 * band}
                        assert clear(doc[i][:n], score[i][:n]) <= set(od.tolist()) and clear(od, os_) <= set(doc[i][:n].tolist())
    assert differs >= 6  # the shortcut really changes answers here
    sh.set_strategy(0)
    # under a field filter the reference switches the shortcut off: the marked bit changes nothing
    qf = sh.make_queries([[0, 1]], S.QueryType.Intersection, field_filter=[0, 2])
    a = sh.search_lexical_batch(qf, 10)
    qm = qf.copy()
    qm["op"][0] |= 0x80000000
    b = sh.search_lexical_batch(qm, 10, reference_shortcuts=False)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    od, os_, otot, _ = O.search_fields_exhaustive(n_docs, dl, boost, offs, D, F, T, [0, 1], O.OP_AND, 10, deleted=gone, field_filter=[0, 2])
    assert int(a[3][0]) == otot and np.allclose(a[1][0][:len(od)], os_, rtol=rel)
    sh.close()
