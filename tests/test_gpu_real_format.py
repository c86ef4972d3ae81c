"""Drop-in rehearsal: shard files in the reference's format (index.bin with >= 1 M keys / three fields, vector.bin, delete.bin) opened, uploaded and queried against the oracle (tools/real_format.py)."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


REL = 1e-4


VOC = [0, 1500, 2500, 3000, 3300, 3600, 3800, 3900, 4000, 4050, 4095]  # df from 0.05 % to 20 %


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
    n_docs = 300_000
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = O.lex_corpus(n_docs, VOC)
    osh = O.Shard(n_docs, dl, offs, docs, tfs)
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, offs, docs, tfs)
    yield sh, osh, n_docs
    sh.close()


def _check_topk(doc, score, cnt, od, os_, abs_tol=0.0):
    """rows sorted desc; scores within REL of the oracle's; identical id sets outside the tie band of the k-th"""
    n = int(cnt)
    assert n == len(od)
    d, s = doc[:n], score[:n]
    assert np.all(s[:-1] >= s[1:])
    assert np.all(doc[n:] == 0xFFFFFFFF)
    assert len(set(map(int, d))) == n
    assert np.allclose(s, os_, rtol=REL, atol=abs_tol)
    if n:
        band = abs(float(os_[-1])) * REL + abs_tol
        clear = lambda dd, ss: {int(x) for x, y in zip(dd, ss) if y > os_[-1] + 2 * band}
        assert clear(d, s) <= {int(x) for x in od} and clear(od, os_) <= {int(x) for x in d}


def _same(a, b, what):
    for x, y, name in zip(a, b, ("doc", "score", "count", "total")):
        assert np.array_equal(x, y), (what, name)


def _oracle_check(S, O, osh, cs, oop, rt, got, k=10):
    doc, score, cnt, tot = got
    for i, (pos, neg) in enumerate(cs):
        od, os_, otot = osh.search_exhaustive(pos, oop, k, not_terms=neg)
        if rt != S.ResultType.Topk:
            assert int(tot[i]) == otot, (pos, neg, rt)
        if rt != S.ResultType.Count:
            _check_topk(doc[i], score[i], cnt[i], od, os_)


def _level_slices(n_docs, offs, docs, tfs, n_terms=None):
    """CSR of a corpus -> per 65 536-doc level (doclen slice bounds, offs, docs, tfs) over the first n_terms terms"""
    nt = len(offs) - 1 if n_terms is None else n_terms
    out = []
    for lv in range((n_docs + 65535) // 65536):
        lo, hi = lv * 65536, min(n_docs, (lv + 1) * 65536)
        lo_, do_, to_ = [0], [], []
        for t in range(nt):
            a, b = int(offs[t]), int(offs[t + 1])
            i0, i1 = a + int(np.searchsorted(docs[a:b], lo)), a + int(np.searchsorted(docs[a:b], hi))
            do_.append(docs[i0:i1]); to_.append(tfs[i0:i1]); lo_.append(lo_[-1] + (i1 - i0))
        out.append((lo, hi, np.asarray(lo_, np.uint64), np.concatenate(do_) if do_ else np.zeros(0, np.uint32),
                    np.concatenate(to_) if to_ else np.zeros(0, np.uint16)))
    return out


def _tiered_shard(S, O, n_docs=150_000, seed=21):
    """a dense image of 5 lists + 9 sparse lists that overlap each other and the dense lists; the oracle holds all 14 as ordinary lists"""
    rng = np.random.default_rng(seed)
    dl = O.lex_doclen(n_docs)
    offs, docs, tfs = [0], [], []
    for df in (0.004, 0.02, 0.07, 0.15, 0.33):
        d = np.sort(rng.choice(n_docs, int(df * n_docs), replace=False)).astype(np.uint32)
        docs.append(d); tfs.append(np.minimum(rng.geometric(0.6, len(d)), 60).astype(np.uint16)); offs.append(offs[-1] + len(d))
    nd = len(offs) - 1
    hot = np.sort(rng.choice(n_docs, 6000, replace=False))
    sp_n = [2, 50, 400, 1500, 3000, 9, 65, 2200, 700]
    s_offs, s_docs, s_tfs = [0], [], []
    for n in sp_n:
        d = np.sort(rng.choice(hot, n, replace=False)).astype(np.uint32)
        s_docs.append(d); s_tfs.append(np.minimum(rng.geometric(0.5, n), 30).astype(np.uint16)); s_offs.append(s_offs[-1] + n)
    d_offs, d_docs, d_tfs = np.asarray(offs, np.uint64), np.concatenate(docs), np.concatenate(tfs)
    s_offs, s_docs, s_tfs = np.asarray(s_offs, np.uint64), np.concatenate(s_docs), np.concatenate(s_tfs)
    sh = S.Shard(0)
    sh.upload_lexical(n_docs, dl, d_offs, d_docs, d_tfs)
    assert sh.append_sparse(s_offs, s_docs, s_tfs) == nd
    osh = O.Shard(n_docs, dl, np.concatenate([d_offs, d_offs[-1] + s_offs[1:]]), np.concatenate([d_docs, s_docs]), np.concatenate([d_tfs, s_tfs]))
    return sh, osh, nd, len(sp_n), hot, n_docs


def _check_against(osh, O, S, out, cases, op, k, rt):
    d, s_, c, t = out
    for i, (terms, nots) in enumerate(cases):
        od, os_, otot = osh.search_exhaustive(terms, op, k, not_terms=nots)
        assert int(t[i]) == otot, (op, k, i, terms, nots, int(t[i]), otot)
        if rt == S.ResultType.Count:
            continue
        assert c[i] == len(od), (op, k, i, terms, nots, int(c[i]), len(od))
        assert np.allclose(s_[i, :c[i]], os_, rtol=1e-4), (op, k, i, terms, nots)
        if len(od) < k:
            assert set(d[i, :c[i]].tolist()) == set(int(x) for x in od)


def _same_answers(a, b):
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_drop_in_rehearsal_on_a_million_doc_index_bin(S, O):
    """index.bin (1 M docs, >= 1 M keys, clustered doc ids, NgramFF | NgramFFF keys, positions) + vector.bin +
    delete.bin as the reference lays them out -> ss_index_bin_open -> tier -> upload with positions; 2-term ANDs, 3-term ORs (rare
    terms from the sparse tier included), phrases over n-gram keys, vector and hybrid queries, every answer against the oracle;
    64 concurrent callers through Index::search of the C++ mirror.  tools/real_format.py holds the rehearsal (bench.py runs it too)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import real_format
    r = real_format.run(n_docs=1_000_000, vocab=1_000_000, n_queries=64, seconds=0.6)
    assert r["files"]["keys"] >= 1_000_000 and r["files"]["ngram_keys"] > 10_000
    assert r["open"]["sparse_terms"] > 900_000 and r["open"]["dense_terms"] > 1000
    assert r["queries"]["phrases_with_ngram_keys"] > 0 and r["queries"]["ors_naming_a_sparse_term"] > 0
    assert r["queries"]["phrases_naming_a_sparse_term"] > 0
    assert set(r["parity"]["queries"]) == {"and2", "or3", "phrase", "vector", "hybrid"}
    assert all(v["errors"] == 0 for v in r["concurrent_callers"].values())


def test_drop_in_rehearsal_on_an_index_bin_with_three_fields(S, O):
    """the same rehearsal over THREE indexed fields (title / body / tags spans; multi-field records and n-gram keys with their
    components' field vectors in the file, BM25F boosts): tiers with the rare keys' merged lists in the sparse tier, positions of both
    tiers, 2-term ANDs (also under a field filter), 3-term ORs, phrases inside one field over keys of either tier, vector + hybrid --
    every answer against the BM25F oracle over the mini indexer's own (doc, field) entries; 64 callers through the C++ mirror"""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import real_format
    r = real_format.run(n_docs=400_000, vocab=400_000, n_queries=48, seconds=0.4, n_fields=3)
    assert r["indexed_fields"] == 3 and r["files"]["ngram_keys"] > 10_000
    assert r["open"]["sparse_terms"] > 300_000 and r["open"]["dense_terms"] > 500
    assert r["queries"]["phrases_with_ngram_keys"] > 0 and r["queries"]["ors_naming_a_sparse_term"] > 0
    assert set(r["parity"]["queries"]) == {"and2", "or3", "and2_body", "phrase", "vector", "hybrid"}
    assert all(v["errors"] == 0 for v in r["concurrent_callers"].values())
