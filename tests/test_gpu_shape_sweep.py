"""The SHAPE SWEEP (VERDICT r5 "next" 1): no query the reference answers may come back empty.

Every cell of   op (Intersection / Union / Phrase)  x  unique terms 1 .. 32  x  NOT terms 0 .. 4  x  indexed fields 1 / 3
                x  field filter  x  dense / sparse tier  x  Topk / TopkCount / Count  (+ the all_terms_frequent shortcut, k up to 300)
is either answered (SS_OK) with oracle parity, or stands in the explicit CPU fall-through list below -- the list INTEGRATION.md
section 4 names; SS_ENOTSUP there means "the crate's own dispatch block (search.rs:3374-3560) answers this one", never "no hits".
The cells beyond the specialised kernels run on the generic galloping kernels (csrc/bm25_gallop.hip) or are composed from the
reference's own sub-queries (ss_api.hip bm25_search_compose); a mixed batch is split per kernel family behind the ABI and comes
back in the callers' order -- which this test exercises by sending every cell of a (world, k, result type) in ONE call.

Oracles (test infrastructure): oracle/ss_oracle.c through oracle.py -- exhaustive BM25 / BM25F with NOT terms, tombstones and field
filters, the all_terms_frequent rule, the phrase loop; for unions under a field filter the brute-force statement of the reference's
sub-query rule (tests/test_gpu_parity.py test_union_under_a_field_filter_follows_the_reference_decomposition)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# the CPU fall-through list: (what, where the reference answers it) -- mirrored by INTEGRATION.md section 4
CPU_FALL_THROUGH = {
    "gt32_terms": "a query of more than 32 unique terms, NOT terms included (union.rs:233-259, 617-624: union_scan_32 over the 32 lists with the largest block maxima + union_count); refused by the mirrors' make_query, tests/test_gpu_union_many.py",
    "union_filter_gt10_no_rows": "a UNION of more than 10 terms under a field filter on a rationed vocabulary (a dense list without a probe row) or an image without "
                                 "merged lists: union.rs:598-805 union_scan_32 + add_result.rs:3124-3136; every other one is answered by that rule behind the ABI",
    "nomerged_phrase": "a phrase on an image of several indexed fields whose boosts kept the merged lists from being built (add_result.rs:3248-3386)",
    "nomerged_frequent": "all_terms_frequent on such an image (add_result.rs:1595-1607)",
    "nomerged_union_lists": "a union of more than 32 (term, field) lists on such an image (union.rs:403-805)",
}


@pytest.fixture(scope="module")
def S():
    import seekstorm_amd
    return seekstorm_amd


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def _check(doc, score, cnt, tot, od, os_, otot, rt, k, S, what):
    if rt != S.ResultType.Topk:
        assert int(tot) == int(otot), (what, "total", int(tot), int(otot))
    if rt == S.ResultType.Count:
        return
    n = min(k, len(od))
    assert int(cnt) == n, (what, "count", int(cnt), n)
    if n == 0:
        return
    assert np.allclose(score[:n], os_[:n], rtol=1e-4, atol=1e-7), (what, "scores", score[:n], os_[:n])
    kth = float(os_[n - 1])
    band = abs(kth) * 2e-4 + 1e-7
    # docs clearly above the k-th score (beyond the tolerance band, where either side may order ties its own way) are the same
    got = {int(d) for d, s in zip(doc[:n], score[:n]) if s > kth + band}
    want = {int(d) for d, s in zip(od[:n], os_[:n]) if s > kth + band}
    assert got <= {int(d) for d in od[:n]} and want <= {int(d) for d in doc[:n]}, (what, "docs above the tie band", sorted(got ^ want)[:8])


def _run(S, sh, queries, k, rt, shortcuts=True):
    return sh.search_lexical_batch(queries, k, rt, reference_shortcuts=shortcuts)


def _expect_enotsup(S, sh, q, k, rt, what):
    from seekstorm_amd import _native as N
    with pytest.raises(N.SeekStormHipError) as e:
        sh.search_lexical_batch(q, k, rt, reference_shortcuts=False)
    assert e.value.code == N.SS_ENOTSUP, (what, e.value.code)


# ------------------------------------------------------------------------------------------------ one indexed field
def _single_field_world(S, O, with_tier):
    """40 K docs: terms 0..13 in 70-80 % of the docs with long tfs (the all_terms_frequent rule holds for them, and enough docs have
    every tf >= 10 to be ranked), 14..37 between 0.5 % and 30 %, 38..49 rare (10 .. 400 postings: the sparse tier when with_tier)"""
    rng = np.random.default_rng(601)
    n_docs = 40_000
    dfs = [int(n_docs * x) for x in np.linspace(0.80, 0.70, 14)] + [int(n_docs * x) for x in np.geomspace(0.30, 0.005, 24)] + \
          [int(x) for x in np.geomspace(400, 10, 12)]
    offs, docs, tfs = [0], [], []
    for t, df in enumerate(dfs):
        d = np.sort(rng.choice(n_docs, df, replace=False)).astype(np.uint32)
        tf = np.minimum(rng.geometric(0.025 if t < 14 else 0.4, df), 300).astype(np.uint16)
        docs.append(d); tfs.append(tf); offs.append(offs[-1] + df)
    offs, docs, tfs = np.asarray(offs, np.uint64), np.concatenate(docs), np.concatenate(tfs)
    dl = O.lex_doclen(n_docs)
    nd = 38 if with_tier else len(dfs)
    sh = S.Shard(0)
    e = int(offs[nd])
    sh.upload_lexical(n_docs, dl, offs[:nd + 1], docs[:e], tfs[:e])
    if with_tier:
        assert sh.append_sparse(offs[nd:] - offs[nd], docs[e:], tfs[e:]) == nd
    osh = O.Shard(n_docs, dl, offs, docs, tfs)
    return sh, osh, n_docs, len(dfs)


def _pick(rng, n, pool):
    return [int(x) for x in rng.choice(pool, n, replace=False)]


@pytest.mark.parametrize("with_tier", [False, True])
def test_sweep_one_indexed_field(S, O, with_tier):
    sh, osh, n_docs, n_terms = _single_field_world(S, O, with_tier)
    rng = np.random.default_rng(17)
    frequent, mid, rare = list(range(0, 14)), list(range(14, 38)), list(range(38, 50))
    cells = []  # (terms, not_terms, op)
    for op in ("and", "or"):
        for n in (1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 16, 24, 28, 32):
            for nn in (0, 1, 4):
                if n + nn > 32:
                    continue
                for tier in (("dense", "mixed") if with_tier else ("dense",)):
                    pool = frequent + mid if tier == "dense" else frequent + mid + rare
                    terms = _pick(rng, n, pool)
                    if tier == "mixed" and not any(t >= 38 for t in terms):
                        terms[-1] = int(rng.choice([r for r in rare if r not in terms]))
                    if op == "and" and n > 4:  # deep intersections stay non-empty as long as they can: the frequent terms first
                        keep = [t for t in terms if t >= 38][:1]
                        terms = _pick(rng, min(n - len(keep), 14), frequent) + keep
                        terms += mid[:n - len(terms)]
                    nots = _pick(rng, nn, [t for t in mid + (rare if tier == "mixed" else []) if t not in terms])
                    cells.append((terms, nots, op))
    # the all_terms_frequent rule over 2 .. 14 frequent terms (marked by the mirror exactly as the reference marks them)
    for n in (2, 3, 7, 8, 9, 12, 14):
        cells.append((_pick(rng, n, frequent), [], "and"))
        cells.append((_pick(rng, n, frequent), _pick(rng, 1, mid), "and"))
    gone = list(range(5, n_docs, 89))
    for deleted in ((), gone):
        sh.set_deleted(deleted)
        osh.set_deleted(deleted)
        for k in (10, 300):
            q = sh.make_queries([c[0] for c in cells], [S.QueryType.Intersection if c[2] == "and" else S.QueryType.Union for c in cells],
                                [c[1] for c in cells])
            if k == 10:  # (k = 300: N <= 256 k, the rule is off -- in the mirror and in the oracle alike)
                n_marked = int(((sh.mark_all_terms_frequent(q, k)["op"] >> 31) & 1).sum())
                assert n_marked >= 14  # the rule holds for the frequent-only intersections (every n: the 7-term limit is gone)
            before = sh.generic_batches()
            for rt in (S.ResultType.TopkCount, S.ResultType.Topk, S.ResultType.Count):
                doc, score, cnt, tot = _run(S, sh, q, k, rt)
                for i, (terms, nots, op) in enumerate(cells):
                    od, os_, otot = osh.search_exhaustive(terms, O.OP_AND if op == "and" else O.OP_OR, k, nots, reference_shortcuts=True)
                    _check(doc[i], score[i], cnt[i], tot[i], od, os_, otot, rt, k, S, ("1f", with_tier, terms, nots, op, k, rt, bool(deleted)))
            assert k != 10 or sh.generic_batches() > before  # the > 7-term all_terms_frequent cells ran on the generic kernel
    # the bit on a query the rule does NOT hold for is ignored (the reference would not have set it)
    sh.set_deleted(())
    osh.set_deleted(())
    q = sh.make_queries([[0, 20], [1, 2, 30]], S.QueryType.Intersection)
    q["op"] |= np.uint32(0x80000000)
    doc, score, cnt, tot = _run(S, sh, q, 10, S.ResultType.TopkCount, shortcuts=False)
    for i, terms in enumerate(([0, 20], [1, 2, 30])):
        od, os_, otot = osh.search_exhaustive(terms, O.OP_AND, 10)
        _check(doc[i], score[i], cnt[i], tot[i], od, os_, otot, S.ResultType.TopkCount, 10, S, ("flag ignored", terms))
    sh.close()


# ------------------------------------------------------------------------------------------------ phrases, one indexed field
def test_sweep_phrases_one_indexed_field(S, O):
    from test_gpu_phrase import _corpus
    n_docs = 30_000
    dfs = [9_000, 8_000, 7_000, 6_500, 6_000, 5_000, 4_500, 4_000, 3_500, 3_000, 2_500, 2_000, 300, 120, 40]
    nd = 12  # 12 .. 14: the sparse tier
    phrases = [[0, 1], [2, 3, 4], [0, 1, 2, 3, 4, 5], [0, 1, 2, 3, 4, 5, 6], [3, 4, 5, 6, 7, 8, 9, 10], [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11],
               [0, 12], [1, 13, 2], [0, 1, 2, 3, 12, 5, 6], [14, 13, 12, 0, 1, 2, 3, 4, 5, 6, 7, 8], [5, 5, 6, 5, 6, 7, 8, 9, 10, 11, 0, 1],
               [4, 3, 2, 1, 0, 11, 10, 9, 8, 7]]
    plant = [(p, 25) for p in phrases]
    dl, offs, docs, tfs, positions = _corpus(O, n_docs, dfs, 41, plant)
    e = int(offs[nd]); pe = int(tfs[:e].astype(np.int64).sum())
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, offs[:nd + 1], docs[:e], tfs[:e], positions[:pe])
    assert sh.append_sparse(offs[nd:] - offs[nd], docs[e:], tfs[e:], positions=positions[pe:]) == nd
    osh = O.Shard(n_docs, dl, offs, docs, tfs)
    osh.set_positions(positions)
    cases = [(p, []) for p in phrases] + [(phrases[3], [11]), (phrases[4], [0, 13]), (phrases[9], [9]), (phrases[1], [12])]
    q = sh.make_queries([c[0] for c in cases], S.QueryType.Phrase, [c[1] for c in cases])
    gone = list(range(3, n_docs, 61))
    for deleted in ((), gone):
        sh.set_deleted(deleted)
        osh.set_deleted(deleted)
        for k in (10, 100, 200):  # 200: k > 128 -- every phrase on the generic kernel
            before = sh.generic_batches()
            for rt in (S.ResultType.TopkCount, S.ResultType.Topk, S.ResultType.Count):
                doc, score, cnt, tot = _run(S, sh, q, k, rt)
                for i, (ph, neg) in enumerate(cases):
                    uniq = list(dict.fromkeys(ph))
                    od, os_, otot = osh.search_phrase(uniq, [uniq.index(w) for w in ph], n_docs)
                    drop = set()
                    for t in neg:
                        drop |= set(docs[int(offs[t]):int(offs[t + 1])].tolist())
                    keep = [j for j, d in enumerate(od.tolist()) if d not in drop]
                    assert len(keep) >= (0 if neg else 10), (ph, neg, len(keep))  # the planted phrases are found
                    _check(doc[i], score[i], cnt[i], tot[i], od[keep], os_[keep], len(keep), rt, k, S, ("phrase 1f", ph, neg, k, rt, bool(deleted)))
            assert sh.generic_batches() > before
    sh.close()


# ------------------------------------------------------------------------------------------------ three indexed fields
def _fields_world(O, n_docs, n_fields, dfs, seed, tf_p):
    rng = np.random.default_rng(seed)
    dl = np.stack([O.lex_doclen(n_docs, seed=O.LEX_SEED + 5 * f) for f in range(n_fields)])
    offs, D, F, T = [0], [], [], []
    for t, df in enumerate(dfs):
        d = np.sort(rng.choice(n_docs, df, replace=False)).astype(np.uint32)
        mask = rng.integers(1, 1 << n_fields, df)
        dd = np.concatenate([d[((mask >> f) & 1) == 1] for f in range(n_fields)])
        ff = np.concatenate([np.full(int((((mask >> f) & 1) == 1).sum()), f, np.uint8) for f in range(n_fields)])
        order = np.lexsort((ff, dd))
        D.append(dd[order]); F.append(ff[order])
        T.append(np.minimum(rng.geometric(tf_p(t), len(dd)), 200).astype(np.uint16))
        offs.append(offs[-1] + len(dd))
    return dl, np.asarray(offs, np.uint64), np.concatenate(D), np.concatenate(F), np.concatenate(T)


def _gated_union_oracle(per_term, terms, nots, filt, gone_set, k):
    """the reference's sub-query rule for a union under a field filter (union.rs:1330-1425 + add_result.rs:3124-3136): the sum over the
    doc's terms that stand in a listed field (all fields of those terms counted); totals: two terms |pass(X) u pass(Y)|, more the
    UNFILTERED union"""
    sc, passing, present = {}, [], set()
    for t in terms:
        ts, dd, ff = per_term[t]
        pas = set(dd[np.isin(ff, list(filt))].tolist()) - gone_set
        passing.append(pas)
        present |= set(dd.tolist()) - gone_set
        for d in pas:
            sc[d] = np.float32(sc.get(d, np.float32(0)) + np.float32(ts[d]))
    banned = set()
    for t in nots:
        banned |= set(per_term[t][1].tolist())
    want = sorted(((d, float(v)) for d, v in sc.items() if d not in banned), key=lambda e: (-e[1], e[0]))[:k]
    total = len((passing[0] | passing[1]) - banned) if len(terms) == 2 else len(present - banned)
    return np.array([w[0] for w in want], np.uint32), np.array([w[1] for w in want], np.float32), total


def _gated_scan_rule_oracle(per_term, terms, nots, filt, gone_set, k):
    """the reference's rule for a union of MORE than 10 terms under a field filter (union_blockid -> union_scan_32, union.rs:598-805, whose
    candidates meet the filter in add_result_multiterm_multifield, add_result.rs:3124-3136: the loop over the doc's present terms returns
    at the first one that stands in no listed field): a doc answers iff EVERY term it holds passes, and scores with all of them;
    union_scan counted it before the filter saw it -- the total is the unfiltered union's"""
    present = {}
    for t in terms:
        ts, dd, ff = per_term[t]
        pas = set(dd[np.isin(ff, list(filt))].tolist())
        for d in set(dd.tolist()) - gone_set:
            e = present.setdefault(d, [True, np.float32(0)])
            e[0] = e[0] and d in pas
            e[1] = np.float32(e[1] + np.float32(ts[d]))
    banned = set()
    for t in nots:
        banned |= set(per_term[t][1].tolist())
    want = sorted(((d, float(e[1])) for d, e in present.items() if e[0] and d not in banned), key=lambda x: (-x[1], x[0]))[:k]
    return np.array([w[0] for w in want], np.uint32), np.array([w[1] for w in want], np.float32), len(set(present) - banned)


@pytest.mark.parametrize("with_tier", [False, True])
def test_sweep_three_indexed_fields(S, O, with_tier):
    n_docs, n_fields, boost = 30_000, 3, [2.0, 1.0, 0.5]
    dfs = [int(n_docs * x) for x in np.linspace(0.78, 0.66, 12)] + [int(n_docs * x) for x in np.geomspace(0.30, 0.01, 22)] + \
          [int(x) for x in np.geomspace(300, 8, 8)]
    dl, offs, docs, fields, tfs = _fields_world(O, n_docs, n_fields, dfs, 88, lambda t: 0.03 if t < 12 else 0.4)
    nd = 34 if with_tier else len(dfs)
    e = int(offs[nd])
    sh = S.Shard(0)
    sh.upload_lexical_fields(n_docs, dl, boost, offs[:nd + 1], docs[:e], fields[:e], tfs[:e])
    assert sh.fields_info()[1]  # merged lists
    if with_tier:
        assert sh.append_sparse_fields(offs[nd:] - offs[nd], docs[e:], fields[e:], tfs[e:]) == nd
    rng = np.random.default_rng(29)
    frequent, mid, rare = list(range(0, 12)), list(range(12, 34)), list(range(34, 42))
    ex = lambda terms, op, k, nots, deleted, filt=(): O.search_fields_exhaustive(n_docs, dl, boost, offs, docs, fields, tfs, terms, op, k, nots, deleted,
                                                                                field_filter=filt)[:3]
    gone = list(range(7, n_docs, 83))
    for deleted in ((), gone):
        sh.set_deleted(deleted)
        gone_set = set(deleted)
        per_term = {}
        for t in range(len(dfs)):
            d, s_, _ = ex([t], O.OP_OR, n_docs, (), deleted)
            a, b = int(offs[t]), int(offs[t + 1])
            per_term[t] = (dict(zip(d.tolist(), s_.tolist())), docs[a:b], fields[a:b])
        for filt in ((), (0,), (1, 2)):
            cells = []
            for op in ("and", "or"):
                for n in (1, 2, 3, 5, 7, 8, 9, 10, 11, 12, 16, 24, 32):
                    for nn in (0, 2):
                        if n + nn > 32:
                            continue
                        for tier in (("dense", "mixed") if with_tier else ("dense",)):
                            pool = frequent + mid + (rare if tier == "mixed" else [])
                            if op == "and" and n > 3:
                                terms = _pick(rng, min(n, 12), frequent)
                                terms += _pick(rng, n - len(terms), mid[:n - len(terms) + 2]) if len(terms) < n else []
                                if tier == "mixed":
                                    terms[-1] = int(rng.choice(rare[:3]))
                            else:
                                terms = _pick(rng, n, pool)
                                if tier == "mixed" and not any(t >= 34 for t in terms):
                                    terms[-1] = int(rng.choice([r for r in rare if r not in terms]))
                            nots = _pick(rng, nn, [t for t in mid[8:] + (rare if tier == "mixed" else []) if t not in terms])
                            cells.append((terms, nots, op))
            for k in (10, 150):
                q = sh.make_queries([c[0] for c in cells], [S.QueryType.Union if c[2] == "or" else S.QueryType.Intersection for c in cells],
                                    [c[1] for c in cells], field_filter=filt)
                before = sh.generic_batches()
                for rt in (S.ResultType.TopkCount, S.ResultType.Topk, S.ResultType.Count):
                    doc, score, cnt, tot = _run(S, sh, q, k, rt, shortcuts=False)
                    for i, (terms, nots, op) in enumerate(cells):
                        what = ("3f", with_tier, terms, nots, op, filt, k, rt, bool(deleted))
                        if op == "or" and filt and len(terms) > 10:  # the reference's other rule: union_scan + the per-doc filter
                            od, os_, otot = _gated_scan_rule_oracle(per_term, terms, nots, filt, gone_set, k)
                        elif op == "or" and filt and len(terms) > 1:
                            od, os_, otot = _gated_union_oracle(per_term, terms, nots, filt, gone_set, k)
                        else:
                            od, os_, otot = ex(terms, O.OP_AND if op != "or" else O.OP_OR, k, nots, deleted, filt if (op != "or" or len(terms) == 1) else ())
                        _check(doc[i], score[i], cnt[i], tot[i], od, os_, otot, rt, k, S, what)
                assert not filt or sh.generic_batches() > before, (filt, k)  # filtered intersections of > 8 terms / > 32 lists
            if not filt:  # all_terms_frequent over the merged lists (off under a field filter, add_result.rs:3116), 2 .. 12 terms
                fr = [_pick(rng, n, frequent) for n in (2, 3, 7, 8, 10, 12)]
                q = sh.make_queries(fr, S.QueryType.Intersection)
                assert int(((sh.mark_all_terms_frequent(q, 10)["op"] >> 31) & 1).sum()) == len(fr)
                before = sh.generic_batches()
                for rt in (S.ResultType.TopkCount, S.ResultType.Count):
                    doc, score, cnt, tot = _run(S, sh, q, 10, rt)
                    for i, terms in enumerate(fr):
                        od, os_, otot = O.search_fields_shortcut(n_docs, dl, boost, offs, docs, fields, tfs, terms, 10, deleted)
                        _check(doc[i], score[i], cnt[i], tot[i], od, os_, otot, rt, 10, S, ("3f frequent", with_tier, terms, rt, bool(deleted)))
                assert sh.generic_batches() > before
    # a union of more than 10 terms under a field filter is answered on both tiers (the cells above), through the mirror's single-query entry too
    sh.set_deleted(())
    for n in (11, 16, 30):
        terms = _pick(rng, n, frequent + mid)
        if with_tier:
            terms[-1] = rare[0]
        ro = sh.search_lexical_shard(terms, S.QueryType.Union, 0, 10, field_filter=(0,))
        assert not ro.cpu_dispatch and len(ro.results) == 10
    # ... a page deeper than SS_MAX_K results is answered (in passes: tests/test_gpu_deep_pages.py)
    q = sh.make_queries([_pick(rng, 2, mid)], S.QueryType.Union)
    assert int(sh.search_lexical_batch(q, 1024, S.ResultType.TopkCount, reference_shortcuts=False)[2][0]) > 0
    assert int(sh.search_lexical_batch(q, 1025, S.ResultType.TopkCount, reference_shortcuts=False)[2][0]) > 0
    assert not sh.search_lexical_shard(_pick(rng, 2, mid), S.QueryType.Union, 1000, 25).cpu_dispatch
    sh.close()


def test_sweep_phrases_three_indexed_fields(S, O):
    from test_gpu_phrase import _corpus_fields
    n_docs, n_fields, boost = 20_000, 3, [2.0, 1.0, 0.5]
    dfs = [6_000, 5_500, 5_000, 4_500, 4_000, 3_500, 3_000, 2_500, 2_000, 1_500, 200, 60]
    nd = 10
    phrases = [[0, 1], [0, 1, 2, 3, 4, 5], [0, 1, 2, 3, 4, 5, 6], [2, 3, 4, 5, 6, 7, 8, 9, 0, 1], [0, 10], [1, 2, 3, 11, 4, 5, 6, 7], [9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 10, 11]]
    plant = [(p, f, 20) for p in phrases for f in (0, 2)]
    dl, offs, docs, fields, tfs, positions = _corpus_fields(O, n_docs, n_fields, dfs, 57, plant, [(0, 1, 40)])
    e = int(offs[nd]); pe = int(tfs[:e].astype(np.int64).sum())
    sh = S.Shard(0)
    sh.upload_lexical_fields(n_docs, dl, boost, offs[:nd + 1], docs[:e], fields[:e], tfs[:e], positions[:pe])
    assert sh.fields_info() == (3, True, True)
    assert sh.append_sparse_fields(offs[nd:] - offs[nd], docs[e:], fields[e:], tfs[e:], positions=positions[pe:]) == nd
    gone = list(range(3, n_docs, 71))
    for deleted in ((), gone):
        sh.set_deleted(deleted)
        for filt in ((), (0,), (1, 2)):
            q = sh.make_queries(phrases, S.QueryType.Phrase, field_filter=filt)
            for k in (10, 150):
                before = sh.generic_batches()
                for rt in (S.ResultType.TopkCount, S.ResultType.Count):
                    doc, score, cnt, tot = _run(S, sh, q, k, rt)
                    for i, ph in enumerate(phrases):
                        uniq = list(dict.fromkeys(ph))
                        od, os_, otot = O.search_fields_phrase(n_docs, dl, boost, offs, docs, fields, tfs, positions, uniq, [uniq.index(w) for w in ph], k,
                                                               deleted, filt)
                        if not filt or 0 in filt:
                            assert otot >= 10, (ph, filt, otot)
                        _check(doc[i], score[i], cnt[i], tot[i], od, os_, otot, rt, k, S, ("phrase 3f", ph, filt, k, rt, bool(deleted)))
                assert sh.generic_batches() > before
    sh.close()


def test_sweep_fields_without_merged_lists(S, O):
    """boosts too far apart for the merged lists' weight code: every query reads (term, field) lists.  Intersections of any size run on the
    generic kernel over the field lists; what stays with the host's dispatch is listed"""
    n_docs, n_fields, boost = 20_000, 3, [4096.0, 1.0, 1.0 / 4096.0]
    dfs = [int(n_docs * x) for x in np.linspace(0.75, 0.6, 10)] + [int(n_docs * x) for x in np.geomspace(0.3, 0.02, 14)]
    dl, offs, docs, fields, tfs = _fields_world(O, n_docs, n_fields, dfs, 99, lambda t: 0.05 if t < 10 else 0.4)
    sh = S.Shard(0)
    sh.upload_lexical_fields(n_docs, dl, boost, offs, docs, fields, tfs)
    assert sh.fields_info()[:2] == (3, False)
    rng = np.random.default_rng(5)
    ex = lambda terms, op, k, nots, filt=(): O.search_fields_exhaustive(n_docs, dl, boost, offs, docs, fields, tfs, terms, op, k, nots, (), field_filter=filt)[:3]
    cells = []
    for n in (1, 2, 4, 8, 9, 10, 12, 16, 20):
        for nn in (0, 2):
            terms = _pick(rng, min(n, 10), range(10)) + (_pick(rng, n - 10, range(10, 24)) if n > 10 else [])
            cells.append((terms, _pick(rng, nn, [t for t in range(10, 24) if t not in terms]), "and"))
    for n in (1, 2, 4, 8, 10):
        cells.append((_pick(rng, n, range(24)), [], "or"))
    for filt in ((), (1,)):
        use = [c for c in cells if not (filt and c[2] == "or" and len(c[0]) > 1)]
        q = sh.make_queries([c[0] for c in use], [S.QueryType.Union if c[2] == "or" else S.QueryType.Intersection for c in use], [c[1] for c in use],
                            field_filter=filt)
        before = sh.generic_batches()
        for rt in (S.ResultType.TopkCount, S.ResultType.Count):
            doc, score, cnt, tot = _run(S, sh, q, 10, rt, shortcuts=False)
            for i, (terms, nots, op) in enumerate(use):
                od, os_, otot = ex(terms, O.OP_OR if op == "or" else O.OP_AND, 10, nots, filt)
                _check(doc[i], score[i], cnt[i], tot[i], od, os_, otot, rt, 10, S, ("3f no merged lists", terms, nots, op, filt, rt))
        assert sh.generic_batches() > before
    # CPU fall-through on such an image
    q = sh.make_queries([_pick(rng, 11, range(24))], S.QueryType.Union)  # 33 (term, field) lists
    _expect_enotsup(S, sh, q, 10, S.ResultType.TopkCount, CPU_FALL_THROUGH["nomerged_union_lists"])
    q = sh.make_queries([[0, 1]], S.QueryType.Intersection)
    q["op"] |= np.uint32(0x80000000)  # the rule holds (both lists in > half of the docs): the several-fields form needs the merged lists
    _expect_enotsup(S, sh, q, 10, S.ResultType.TopkCount, CPU_FALL_THROUGH["nomerged_frequent"])
    q = sh.make_queries([[0, 1]], S.QueryType.Phrase)
    _expect_enotsup(S, sh, q, 10, S.ResultType.TopkCount, CPU_FALL_THROUGH["nomerged_phrase"])
    sh.close()
