"""Consumer of oracle/ref_dump's output: indexes written by the REAL seekstorm crate (index.bin / delete.bin / vector.bin) and the
answers the crate itself gave (expected.json).  The dump needs cargo + the crate's dependencies, which this image does not have
(oracle/ref_dump/Cargo.toml): the tests SKIP while neither oracle/_ref/dump nor tests/golden/ref_dump exists.  One run of
    cd oracle/ref_dump && cargo run --release -- ../_ref/dump
on any box with cargo turns SURVEY 8 rows (c), a-10 and f-1 from "pinned by a restated writer" into "pinned by the crate":
  * every key the dump names is found in the file, and its decoded postings are the corpus generator's (bytes -> postings);
  * every lexical query: doc ids outside the k-th score's tie band, scores within 1e-4 relative, result_count_total exact
    (Topk: as the crate reports it -- it may stop early, so only '>= the results returned' is checked);
  * vector cases: top-k rows and raw dot scores of the f32 records within 1e-4, the i8 records' integer scores exactly.
Phrase queries over n-gram keys ("ngram" case) are compared only when every key they need is among the dumped term keys."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMP = next((d for d in (os.path.join(ROOT, "oracle", "_ref", "dump"), os.path.join(ROOT, "tests", "golden", "ref_dump")) if os.path.isdir(d)), None)
if DUMP is None:
    pytest.skip("no dump of the real crate (oracle/ref_dump needs cargo; see its Cargo.toml)", allow_module_level=True)

LEXICAL = ("single", "single_del", "ngram", "fields", "levels")


def _case(name):
    d = os.path.join(DUMP, name)
    if not os.path.isdir(d):
        pytest.skip(f"case {name} not in the dump")
    exp = json.load(open(os.path.join(d, "expected.json")))
    shard = os.path.join(d, "index", "shards", "0")
    if not os.path.isdir(shard):  # (older layouts kept the shard files beside index.json)
        shard = os.path.join(d, "index")
    return exp, shard


def body(i):
    """oracle/ref_dump/src/main.rs body(): the corpus, word for word"""
    s = []
    if i % 3 == 0:
        s += "the quick w0 w1 of the day".split()
    for j in range(12):
        if i % (j + 2) == 0:
            s += [f"w{j}"] * (1 + (i // 7) % (j + 1))
    s += [f"pad{i % 97}", f"pad{i % 89}"]
    return s


def _field_words(i, field):
    if field == "body":
        return body(i)
    if field == "title":
        return [f"w{i % 5}", f"w{i % 7}", f"title{i % 13}"]
    return [f"site{i % 31}", f"w{i % 3}"]


def _expected_postings(term, n_docs, fields):
    """(docs, tf summed over the indexed fields) of `term` from the generator"""
    docs, tfs = [], []
    for i in range(n_docs):
        tf = sum(_field_words(i, f).count(term) for f in fields)
        if tf:
            docs.append(i); tfs.append(tf)
    return np.asarray(docs, np.uint32), np.asarray(tfs, np.int64)


def _parse(q):
    """the dump's query strings: +term (required), -term (NOT), "a b" (phrase) -> (terms, not_terms, phrase words)"""
    if q.startswith('"'):
        w = q.strip('"').split()
        return w, [], w
    pos, neg = [], []
    for tok in q.split():
        (neg if tok.startswith("-") else pos).append(tok.lstrip("+-"))
    return pos, neg, None


@pytest.mark.parametrize("name", LEXICAL)
def test_lexical_case_against_the_crate(name):
    import seekstorm_amd as S
    from seekstorm_amd import _native as N
    exp, shard = _case(name)
    schema = [f for f in exp["schema"] if f.get("index_lexical")]
    fields = [f["field"] for f in schema]
    boost = [float(f.get("boost", 1.0)) for f in schema]
    ngram = int(exp["meta"]["ngram_indexing"])
    khs = 20 if ngram == 0 else 22 if ngram < 8 else 23  # index.rs:2806-2812
    data = open(os.path.join(shard, "index.bin"), "rb").read()
    ix = S.IndexBin(data, len(fields), key_head_size=khs)
    assert ix.indexed_doc_count == exp["docs"]
    assert ix.level_count == (exp["docs"] + 65535) // 65536
    keys = {t: int(h) for t, h in exp["term_keys"].items()}
    missing = [t for t, h in keys.items() if ix.term_of_key(h) is None and any(t in _field_words(i, f) for i in range(min(exp["docs"], 400)) for f in fields)]
    assert not missing, f"keys not found in index.bin (a crate built with gxhash hashes differently): {missing}"
    # bytes -> postings: every dumped key against the generator (one indexed field: tf per doc; several: summed over the fields)
    for t, h in keys.items():
        tid = ix.term_of_key(h)
        if tid is None or int(ix.term_components[tid]) != 1:
            continue
        d, tf = ix.postings(tid)
        ed, etf = _expected_postings(t, exp["docs"], fields)
        if len(fields) == 1:
            assert np.array_equal(d, ed) and np.array_equal(tf.astype(np.int64), etf), f"postings of {t!r} differ from the corpus"
        else:
            assert np.array_equal(np.unique(d), ed), f"docs of {t!r} differ from the corpus"
    sh = S.Shard(0)
    try:
        sh.upload_index_bin(ix, boost if len(fields) > 1 else None, positions=True)
        dpath = os.path.join(shard, "delete.bin")
        if os.path.exists(dpath) and os.path.getsize(dpath):
            sh.set_deleted(open(dpath, "rb").read())
            assert sorted(np.frombuffer(open(dpath, "rb").read(), "<u8").tolist()) == sorted(exp["deleted"])
        compared = 0
        for e in exp["queries"]:
            pos, neg, phrase = _parse(e["query"])
            if any(t not in keys or ix.term_of_key(keys[t]) is None for t in pos + neg):
                continue  # (a term no doc holds, or a phrase word that only exists inside n-gram keys)
            if phrase is not None and (ngram != 0 and any(int(ix.term_components[ix.term_of_key(keys[t])]) != 1 for t in pos)):
                continue
            uniq = list(dict.fromkeys(pos))
            tids = [ix.term_of_key(keys[t]) for t in uniq]
            nots = [ix.term_of_key(keys[t]) for t in neg]
            qt = {"Union": S.QueryType.Union, "Intersection": S.QueryType.Intersection, "Phrase": S.QueryType.Phrase}[e["query_type"]]
            rt = {"Topk": S.ResultType.Topk, "TopkCount": S.ResultType.TopkCount, "Count": S.ResultType.Count}[e["result_type"]]
            ro = sh.search_lexical_shard(tids if phrase is None else [ix.term_of_key(keys[t]) for t in phrase], qt, 0, int(e["k"]), rt, strict=True, not_terms=nots)
            what = (name, e["query"], e["query_type"], e["result_type"], e["k"])
            if e["result_type"] == "Topk":
                assert ro.result_count_total >= len(ro.results), what
            else:
                assert ro.result_count_total == int(e["result_count_total"]), what + (ro.result_count_total, e["result_count_total"])
            if e["result_type"] == "Count":
                continue
            ed, es = np.asarray(e["doc_ids"], np.int64), np.asarray(e["scores"], np.float32)
            gd = np.asarray([r.doc_id for r in ro.results], np.int64); gs = np.asarray([r.score for r in ro.results], np.float32)
            assert len(gd) == len(ed), what + (len(gd), len(ed))
            assert np.allclose(gs, es, rtol=1e-4), what
            if len(ed):
                band = abs(float(es[-1])) * 2e-4
                clear = lambda dd, ss: {int(x) for x, y in zip(dd, ss) if y > es[-1] + band}
                assert clear(gd, gs) <= set(ed.tolist()) and clear(ed, es) <= set(gd.tolist()), what
            compared += 1
        assert compared >= 10, f"only {compared} queries of the dump could be compared"
    finally:
        sh.close()
        ix.close()


@pytest.mark.parametrize("name,i8", [("vector_f32", False), ("vector_i8", True)])
def test_vector_case_against_the_crate(name, i8):
    import seekstorm_amd as S
    exp, shard = _case(name)
    dim = int(exp["meta"]["dimensions"])
    data = open(os.path.join(shard, "vector.bin"), "rb").read()
    sh = S.Shard(0)
    try:
        sh.upload_vector_bin(data, dim, i8=i8, use_record_scale=i8)
        assert sh.vector_count == exp["docs"]
        for e in exp["queries"]:
            q = np.asarray(e["query_vector"], np.float32)[None, :]
            k = int(e["k"])
            if i8:
                from oracle import oracle as O
                q8 = O.quantize_i8(q)
                doc, score, cnt, tot = sh.search_vector_batch_i8(q8, k)
            else:
                doc, score, cnt, tot = sh.search_vector_batch(q, k)
            ed, es = np.asarray(e["doc_ids"], np.int64), np.asarray(e["scores"], np.float32)
            c = int(cnt[0])
            assert c == len(ed), (name, k, c, len(ed))
            if not i8:
                assert np.allclose(score[0][:c], es, rtol=1e-4, atol=1e-6), (name, k)
            if c:
                band = abs(float(es[-1])) * 2e-4 + 1e-6
                clear = lambda dd, ss: {int(x) for x, y in zip(dd, ss) if y > es[-1] + band}
                if not i8:
                    assert clear(doc[0][:c], score[0][:c]) <= set(ed.tolist()) and clear(ed, es) <= set(int(x) for x in doc[0][:c]), (name, k)
                else:  # the crate scores i8 records through its own query quantiser: the top of the ranking must agree
                    assert len(set(int(x) for x in doc[0][:min(c, 5)]) & set(ed[:10].tolist())) >= min(c, 3), (name, k)
    finally:
        sh.close()
