"""Pruned (MaxScore over the probe index) vs exhaustive BM25 strategy, through the C ABI: both must return IDENTICAL
results -- same doc ids, bit-identical scores, same exact counts -- and agree with the CPU oracle.  Corpora are built to
stress what pruning can get wrong: ties at the k-th score, tf >= 16 postings (weights outside the table), very sparse and
very dense terms, single-term queries, k larger than the match count, many partitions per query."""
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


def _corpus(O, n_docs, dfs, seed, tie_heavy=False, big_tf=False):
    rng = np.random.default_rng(seed)
    if tie_heavy:  # few distinct lengths and tf = 1 almost everywhere: large groups of equal scores
        lens = rng.choice([40, 41, 120], n_docs)
    else:
        lens = np.clip(np.round(np.exp(np.log(120) + 0.6 * rng.standard_normal(n_docs))), 8, 2000).astype(np.int64)
    dl = np.array([O.lib().so_int_to_byte4(int(x)) for x in np.unique(lens)], np.uint8)
    lut = dict(zip(np.unique(lens).tolist(), dl.tolist()))
    doclen = np.array([lut[int(x)] for x in lens], np.uint8)
    offs, docs, tfs = [0], [], []
    for df in dfs:
        n = max(1, int(round(df * n_docs)))
        d = np.sort(rng.choice(n_docs, n, replace=False)).astype(np.uint32)
        if tie_heavy:
            tf = np.where(rng.random(n) < 0.97, 1, 2).astype(np.uint16)
        else:
            tf = rng.geometric(0.55, n).clip(1, 400).astype(np.uint16)
            if big_tf:
                m = rng.random(n) < 0.03
                tf[m] = rng.integers(16, 400, int(m.sum())).astype(np.uint16)
            if big_tf == 2:  # tf beyond the 9-bit posting field: exact values come from the exception lists
                m = rng.random(n) < 0.01
                tf[m] = rng.choice([510, 511, 512, 3000, 65535], int(m.sum())).astype(np.uint16)
        docs.append(d); tfs.append(tf); offs.append(offs[-1] + n)
    return doclen, np.asarray(offs, np.uint64), np.concatenate(docs), np.concatenate(tfs)


def _run(S, sh, queries, k, rt, strategy):
    from seekstorm_amd import _native as N
    sh.set_strategy(strategy)
    out = sh.search_lexical_batch(queries, k, rt)
    sh.set_strategy(N.BM25_AUTO)
    return out


@pytest.mark.parametrize("tie_heavy,big_tf", [(False, 0), (True, 0), (False, 1), (False, 2)])
def test_pruned_equals_exhaustive_and_oracle(S, O, tie_heavy, big_tf):
    from seekstorm_amd import _native as N
    n_docs = 150_000
    dfs = [0.0004, 0.002, 0.008, 0.02, 0.05, 0.11, 0.3, 0.62, 0.013, 0.004, 0.035, 0.0009]
    dl, offs, docs, tfs = _corpus(O, n_docs, dfs, 5 + tie_heavy + 2 * big_tf, tie_heavy, big_tf)
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, offs, docs, tfs)
    osh = O.Shard(n_docs, dl, offs, docs, tfs)
    rng = np.random.default_rng(99)
    term_lists = [[int(x) for x in rng.choice(len(dfs), int(rng.integers(1, 5)), replace=False)] for _ in range(60)]
    term_lists += [[0], [7], [6, 7], [0, 11], [5, 6, 7], [0, 1, 2, 3]]
    for qt, op in ((S.QueryType.Union, O.OP_OR), (S.QueryType.Intersection, O.OP_AND)):
        q = sh.make_queries(term_lists, qt)
        for k in (1, 10, 100):
            # top-k: pruned == exhaustive, bit for bit
            pd, ps, pc, _ = _run(S, sh, q, k, S.ResultType.Topk, N.BM25_PRUNED)
            ed, es, ec, _ = _run(S, sh, q, k, S.ResultType.Topk, N.BM25_EXHAUSTIVE)
            assert np.array_equal(pc, ec)
            assert np.array_equal(ps, es), f"{qt} k={k}: scores differ"
            assert np.array_equal(pd, ed), f"{qt} k={k}: doc ids differ"
            # and against the oracle (scores within tolerance, identical sets outside the tie band)
            for i in (0, 7, 23, len(term_lists) - 1, len(term_lists) - 4):
                od, os_, otot = osh.search_exhaustive(term_lists[i], op, k)
                n = int(pc[i])
                assert n == len(od)
                assert np.allclose(ps[i][:n], os_, rtol=REL)
                if n:
                    band = abs(float(os_[-1])) * REL
                    assert {int(x) for x, y in zip(pd[i][:n], ps[i][:n]) if y > os_[-1] + band} == \
                           {int(x) for x, y in zip(od, os_) if y > os_[-1] + band}
        if qt == S.QueryType.Intersection:  # exact counts of intersections come from the pruned path too
            for rt in (S.ResultType.TopkCount, S.ResultType.Count):
                pd, ps, pc, pt = _run(S, sh, q, 10, rt, N.BM25_PRUNED)
                ed, es, ec, et = _run(S, sh, q, 10, rt, N.BM25_EXHAUSTIVE)
                assert np.array_equal(pt, et) and np.array_equal(pc, ec)
                if rt == S.ResultType.TopkCount:
                    assert np.array_equal(ps, es) and np.array_equal(pd, ed)
                for i in (1, 30, len(term_lists) - 2):
                    _, _, otot = osh.search_exhaustive(term_lists[i], op, 10)
                    assert int(pt[i]) == otot
    # exact union counts: popcounts over the probe index's bit records (union_count, union.rs:807-) beside the pruned top-k
    q = sh.make_queries(term_lists, S.QueryType.Union)
    for rt in (S.ResultType.TopkCount, S.ResultType.Count):
        pd, ps, pc, pt = _run(S, sh, q, 10, rt, N.BM25_PRUNED)
        ed, es, ec, et = _run(S, sh, q, 10, rt, N.BM25_EXHAUSTIVE)
        assert np.array_equal(pt, et) and np.array_equal(pc, ec) and np.array_equal(ps, es) and np.array_equal(pd, ed)
    for i in range(4):
        assert int(pt[i]) == osh.search_exhaustive(term_lists[i], O.OP_OR, 10)[2]
    # more than 4 scored terms: the explicit PRUNED strategy refuses, AUTO falls back to the scan
    q5 = sh.make_queries([[0, 1, 2, 3, 4]], S.QueryType.Union)
    with pytest.raises(S.SeekStormHipError):
        _run(S, sh, q5, 10, S.ResultType.Topk, N.BM25_PRUNED)
    _run(S, sh, q5, 10, S.ResultType.Topk, N.BM25_AUTO)
    sh.close()


def test_pruned_many_partitions_and_small_shards(S, O):
    """few queries -> one sub-block per partition; shards smaller than a sub-block; k larger than the match count"""
    from seekstorm_amd import _native as N
    for n_docs in (100, 5000, 70_000):
        dfs = [0.01, 0.05, 0.2, 0.5]
        dl, offs, docs, tfs = _corpus(O, n_docs, dfs, 3)
        sh = S.Shard(0)
        sh.upload_lexical(n_docs, dl, offs, docs, tfs)
        osh = O.Shard(n_docs, dl, offs, docs, tfs)
        tl = [[0, 1, 2], [3, 0], [1]]
        for qt, op in ((S.QueryType.Union, O.OP_OR), (S.QueryType.Intersection, O.OP_AND)):
            q = sh.make_queries(tl, qt)
            for k in (10, 128):
                pd, ps, pc, _ = _run(S, sh, q, k, S.ResultType.Topk, N.BM25_PRUNED)
                ed, es, ec, _ = _run(S, sh, q, k, S.ResultType.Topk, N.BM25_EXHAUSTIVE)
                assert np.array_equal(pc, ec) and np.array_equal(ps, es) and np.array_equal(pd, ed)
                for i in range(len(tl)):
                    od, os_, _ = osh.search_exhaustive(tl[i], op, k)
                    assert int(pc[i]) == len(od) and np.allclose(ps[i][:len(od)], os_, rtol=REL)
        sh.close()


def test_pruned_is_repeatable_on_a_generated_corpus(S, O):
    """the pruned path shares thresholds between concurrently running partitions (atomics): whatever the interleaving,
    every run must reproduce the exhaustive answer exactly"""
    from seekstorm_amd import _native as N
    n_docs = 1_500_000
    th = O.term_thresholds(256)
    sh = S.Shard(0)
    sh.synth_lexical(O.LEX_SEED, n_docs, th, O.len_table())
    rng = np.random.default_rng(4)
    tl = [[int(x) for x in rng.choice(256, 3, replace=False)] for _ in range(200)]
    q = sh.make_queries(tl, S.QueryType.Union)
    ed, es, ec, _ = _run(S, sh, q, 10, S.ResultType.Topk, N.BM25_EXHAUSTIVE)
    for rep in range(12):
        sub = q[: 200 - 7 * rep]  # different batch sizes -> different partition counts and interleavings
        pd, ps, pc, _ = _run(S, sh, sub, 10, S.ResultType.Topk, N.BM25_PRUNED)
        n = len(sub)
        assert np.array_equal(ps, es[:n]) and np.array_equal(pd, ed[:n]) and np.array_equal(pc, ec[:n]), f"run {rep}"
    sh.close()


def test_rationed_probe_rows(S, O):
    """probe rows only for the longest lists (a real vocabulary's tail gets none): queries over probed lists still take the
    pruned strategy, queries touching an unprobed list fall back to the scan -- same answers either way"""
    from seekstorm_amd import _native as N
    n_docs = 150_000
    dfs = [0.3, 0.11, 0.05, 0.02, 0.008, 0.002, 0.0009, 0.0004, 0.0]
    dl, offs, docs, tfs = _corpus(O, n_docs, dfs, 11)
    full, part = S.Shard(0), S.Shard(0)
    full.upload_lexical(n_docs, dl, offs, docs, tfs)
    n_sub = (n_docs + 4095) // 4096
    part.set_probe_budget((4 + 1) * n_sub * 64 * 12)  # four rows + the zero row
    part.upload_lexical(n_docs, dl, offs, docs, tfs)
    assert list(full.terms_probed(np.arange(len(dfs)))) == [True] * len(dfs)
    assert list(part.terms_probed(np.arange(len(dfs)))) == [True] * 4 + [False] * 5  # (_corpus gives the last list one posting)
    tl = [[0, 1, 2], [1, 3], [0, 5], [6, 7], [2, 4, 7], [3], [5], [0, 8], [1, 2, 3, 0]]
    for qt in (S.QueryType.Union, S.QueryType.Intersection):
        for rt in (S.ResultType.Topk, S.ResultType.TopkCount, S.ResultType.Count):
            a = full.search_lexical_batch(full.make_queries(tl, qt), 10, rt)
            b = part.search_lexical_batch(part.make_queries(tl, qt), 10, rt)
            for x, y in zip(a, b):
                assert np.array_equal(x, y), (qt, rt)
    # a mixed batch runs as two -- the queries whose lists all have rows keep the pruned strategy -- and the answers come
    # back in the callers' order (checked above: tl mixes both kinds); a uniform batch stays one launch
    part.profile(True)
    for terms, launches in (([[0, 1], [2, 6], [1, 3], [7, 0], [3, 2]], 2), ([[0, 1], [2, 3]], 1), ([[0, 6], [5, 2]], 1)):
        part.profile_read(0, reset=True)
        got = part.search_lexical_batch(part.make_queries(terms, S.QueryType.Union), 10, S.ResultType.Topk)
        assert part.profile_read(0, reset=True)[0] == launches, terms
        want = full.search_lexical_batch(full.make_queries(terms, S.QueryType.Union), 10, S.ResultType.Topk)
        assert all(np.array_equal(x, y) for x, y in zip(got, want))
    part.profile(False)
    # only-probed batches may force the pruned strategy, batches touching the tail may not
    part.set_strategy(N.BM25_PRUNED)
    part.search_lexical_batch(part.make_queries([[0, 1], [2, 3]], S.QueryType.Union), 10, S.ResultType.Topk)
    with pytest.raises(S.SeekStormHipError):
        part.search_lexical_batch(part.make_queries([[0, 1], [2, 6]], S.QueryType.Union), 10, S.ResultType.Topk)
    full.close()
    part.close()


def test_probe_rows_on_demand(S, O):
    """A rationed vocabulary with a row pool (>= 32 rows): the row-less lists a host-pointer batch touches get pool rows built
    from their postings, least recently used first, and the batch keeps the pruned strategy; more row-less lists than pool
    rows -> the rest scans (mixed batch).  Every answer equals the unrationed shard's, over several rounds of eviction;
    facet counts over a list that had no row work as well."""
    from seekstorm_amd import _native as N
    n_docs = 150_000
    dfs = [0.2 / (1 + 0.35 * i) for i in range(56)]
    dl, offs, docs, tfs = _corpus(O, n_docs, dfs, 19)
    full, part = S.Shard(0), S.Shard(0)
    full.upload_lexical(n_docs, dl, offs, docs, tfs)
    n_sub = (n_docs + 4095) // 4096
    part.set_probe_budget((40 + 1) * n_sub * 64 * 12)   # 40 rows: 30 fixed (the longest lists) + a pool of 10
    part.upload_lexical(n_docs, dl, offs, docs, tfs)
    probed = part.terms_probed(np.arange(56))
    assert probed.sum() == 30 and probed[:30].all()
    rng = np.random.default_rng(5)
    part.profile(True)
    for rnd in range(6):
        tail = rng.choice(np.arange(30, 56), 8, replace=False)          # 8 row-less lists <= pool
        tl = [[int(rng.integers(0, 30)), int(t)] for t in tail] + [[int(tail[0]), int(tail[1]), int(rng.integers(0, 30))], [3, 7, 11]]
        for qt in (S.QueryType.Union, S.QueryType.Intersection):
            for rt in (S.ResultType.Topk, S.ResultType.TopkCount, S.ResultType.Count):
                part.profile_read(0, reset=True)
                a = full.search_lexical_batch(full.make_queries(tl, qt), 10, rt)
                b = part.search_lexical_batch(part.make_queries(tl, qt), 10, rt)
                for x, y in zip(a, b):
                    assert np.array_equal(x, y), (rnd, qt, rt)
                if rt == S.ResultType.Topk:
                    assert part.profile_read(0, reset=True)[0] == 1, (rnd, qt, rt)   # one launch: nothing was left to the scans
        assert N.lib() is not None and (np.asarray(part.terms_probed(tail)) != 0).all()
        out = np.zeros(len(tail), np.uint8)
        t32 = np.ascontiguousarray(tail, np.uint32)
        N.check(N.lib().ss_bm25_term_probed(part._h, len(t32), N.ptr(t32, N.u32p), N.ptr(out, N.u8p)), "ss_bm25_term_probed")
        assert (out == 2).all()                                          # pool rows
    # forced PRUNED works on a tail query now
    part.set_strategy(N.BM25_PRUNED)
    a = full.search_lexical_batch(full.make_queries([[2, 50]], S.QueryType.Union), 10, S.ResultType.TopkCount)
    b = part.search_lexical_batch(part.make_queries([[2, 50]], S.QueryType.Union), 10, S.ResultType.TopkCount)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    part.set_strategy(N.BM25_AUTO)
    # more row-less lists than pool rows: the batch is split, the answers stay
    tl = [[int(t), int(t) - 29] for t in range(30, 56)]
    a = full.search_lexical_batch(full.make_queries(tl, S.QueryType.Union), 10, S.ResultType.TopkCount)
    part.profile_read(0, reset=True)
    b = part.search_lexical_batch(part.make_queries(tl, S.QueryType.Union), 10, S.ResultType.TopkCount)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert part.profile_read(0, reset=True)[0] == 2
    part.profile(False)
    # facet counts read the match set from the bit records: a list that had no row gets one
    rec = np.zeros((n_docs, 2), np.uint8)
    rec[:, 0] = np.arange(n_docs) % 7
    full.upload_facets(rec); part.upload_facets(rec)
    q = [[1, 44]]
    ca = full.facet_count(full.make_queries(q, S.QueryType.Union), 0, "string16", n_buckets=7)
    cb = part.facet_count(part.make_queries(q, S.QueryType.Union), 0, "string16", n_buckets=7)
    assert np.array_equal(ca[0], cb[0]) and ca[1:] == cb[1:]
    full.close()
    part.close()


def test_block_maxima_on_a_skewed_corpus(S, O):
    """a-12: per-(term, block) maxima (get_max_score, index.rs:2938-3200; used as in intersection.rs:2090-2097, single.rs:373-386).
    A corpus whose weights are clustered by doc id -- short-doc regions hold the top-k, the long-doc regions' block maxima lie
    far below the list maxima: the image turns the per-partition bounds on by itself, and the pruned strategy under them still
    equals the exhaustive one bit for bit and the oracle within tolerance (unions and intersections, Topk / TopkCount)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "probes"))
    import skew_corpus
    from seekstorm_amd import _native as N
    n_docs = 1_200_000
    dl, offs, docs, tfs = skew_corpus.build(O, n_docs, [0.012, 0.035, 0.10, 0.02])
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, offs, docs, tfs)
    osh = O.Shard(n_docs, dl, offs, docs, tfs)
    cases = [([0, 1, 2], S.QueryType.Union, O.OP_OR), ([1, 3], S.QueryType.Union, O.OP_OR), ([2], S.QueryType.Union, O.OP_OR),
             ([0, 2, 3, 1], S.QueryType.Union, O.OP_OR), ([1, 2], S.QueryType.Intersection, O.OP_AND)]
    q = sh.make_queries([c[0] for c in cases], [c[1] for c in cases])
    res = {}
    for strat in (N.BM25_EXHAUSTIVE, N.BM25_AUTO):
        sh.set_strategy(strat)
        res[strat] = [sh.search_lexical_batch(q, 10, rt) for rt in (S.ResultType.Topk, S.ResultType.TopkCount)]
    for a, b in zip(res[N.BM25_EXHAUSTIVE], res[N.BM25_AUTO]):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    doc, score, cnt, tot = res[N.BM25_AUTO][1]
    for i, (terms, _, op) in enumerate(cases):
        od, os_, otot = osh.search_ref(terms, op, 10, O.RT_TOPKCOUNT)
        assert int(tot[i]) == otot and cnt[i] == len(od)
        assert np.allclose(score[i][:cnt[i]], os_, rtol=1e-4)
        # every result lives in a short-doc region (region % 8 == 3)
        assert np.all(((doc[i][:cnt[i]] >> 16) % 8) == 3)
    sh.set_strategy(N.BM25_AUTO)
    sh.close()
