"""Drop-in rehearsal at configuration size: a shard's files as the reference writes them -- index.bin of a
text-shaped corpus (Zipf vocabulary, clustered doc ids, NgramFF | NgramFFF keys, positions; oracle/ss_textindex.c), vector.bin,
delete.bin -- opened through ss_index_bin_open -> ss_index_bin_tier -> ss_bm25_upload_index_bin_positions, then C1-shaped 2-term ANDs,
3-term ORs, phrases (n-gram keys as the query tokenizer resolves them) and hybrid queries, every answer checked against the oracle on
the corpus' own lists; concurrent single-query callers through Index::search of the C++ mirror.

Shared by tests/test_gpu_round4.py and bench.py (`real_format` leg).  Not part of the product: it drives the product and the checker.
"""
import ctypes as C
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def vector_bin(rows, dim):
    """vector.bin (vector.rs:1066-1094) with Clustering::None: per 65 536-doc level one cluster holding every record of the level;
    record = packed VectorHeader (u16 doc_id, u32 field_id, u32 chunk_id, f32 scale, f32 norm, i16 zero_point, i32 sum_q) + dim f32"""
    n = len(rows)
    rec = np.dtype([("doc_id", "<u2"), ("field_id", "<u4"), ("chunk_id", "<u4"), ("scale", "<f4"), ("norm", "<f4"), ("zp", "<i2"), ("sum_q", "<i4"),
                    ("v", "<f4", (dim,))])
    assert rec.itemsize == 24 + 4 * dim
    out = bytearray()
    for l0 in range(0, n, 65536):
        m = min(65536, n - l0)
        out += (1).to_bytes(4, "little") + int(m).to_bytes(4, "little")
        a = np.zeros(m, rec)
        a["doc_id"] = np.arange(m, dtype=np.uint16)
        a["scale"], a["norm"] = 1.0, 1.0
        a["v"] = rows[l0:l0 + m]
        out += a.tobytes()
    return bytes(out)


def run(n_docs=1_000_000, vocab=1_000_000, dim=64, n_queries=96, dense_min=2000, k=10, seed=11, parity=True, callers=64, seconds=1.0, log=None,
        n_fields=1, boost=None, probe=False):
    """n_fields > 1: the same rehearsal over an index with several indexed fields (title / body / tags spans of every doc; BM25F with
    `boost`, multi-field records and n-gram keys in the file, the sparse tier's merged lists, phrases inside one field, field-filtered ANDs)"""
    import seekstorm_amd as S
    from seekstorm_amd import _native as N
    from seekstorm_amd.search import idf_f32
    from oracle import oracle as O
    from oracle import fullsize as F
    from oracle import textindex as TI
    say = log or (lambda *a: None)
    out = {"docs": n_docs, "vocabulary": vocab}
    # ---- the files (test infrastructure: the mini indexer)
    t0 = time.perf_counter()
    MF = n_fields > 1
    if MF and boost is None:
        boost = [2.0, 1.0, 0.5, 0.25][:n_fields]
    T = TI.TextCorpus(seed, n_docs, vocab, n_frequent=64, mean_len=100.0, topic_share=0.35, n_fields=n_fields, longest_field=1 if MF else 0)
    data = T.write_index_bin(key_head_size=23)
    out["indexed_fields"] = n_fields
    rng = np.random.default_rng(seed)
    rows = O.vec_gen(O.VEC_SEED, 0, n_docs, dim)
    vbin = vector_bin(rows, dim)
    gone = np.unique(rng.integers(0, n_docs, n_docs // 200, dtype=np.uint64))
    dbin = gone.astype("<u8").tobytes()  # delete.bin: a stream of u64 (index.rs:3798-3809)
    out["files"] = {"index_bin_bytes": len(data), "vector_bin_bytes": len(vbin), "delete_bin_bytes": len(dbin), "tokens": T.n_tokens, "keys": T.n_keys_nonempty,
                    "ngram_keys": T.n_ngram_keys, "postings": T.n_postings, "write_s": time.perf_counter() - t0}
    say("files", out["files"])
    # ---- open: walk, tier, decode, upload (what open_shard's end would call)
    sh = S.Shard(0)
    t0 = time.perf_counter()
    ix = S.IndexBin(data, n_fields, key_head_size=23)
    t_open = time.perf_counter() - t0
    t0 = time.perf_counter()
    n_dense = ix.tier(dense_min)
    t_tier = time.perf_counter() - t0
    t0 = time.perf_counter()
    sh.upload_index_bin(ix, boost, positions=True)
    t_up = time.perf_counter() - t0
    t0 = time.perf_counter()
    sh.upload_vector_bin(vbin, dim)
    sh.set_deleted(dbin)
    t_vec = time.perf_counter() - t0
    info = sh.lexical_info()
    n_sp, p_sp, b_sp = sh.sparse_info()
    n_post_all = int(info["n_postings"]) + p_sp
    out["open"] = {"index_bin_open_s": t_open, "tier_s": t_tier, "decode_and_upload_s": t_up, "vector_bin_and_delete_bin_s": t_vec,
                   "open_s": t_open + t_tier + t_up + t_vec, "terms": ix.term_count, "dense_terms": n_dense, "sparse_terms": n_sp,
                   "dense_postings": int(info["n_postings"]), "sparse_postings": p_sp, "sparse_tier_bytes": b_sp,
                   "postings_per_s": n_post_all / max(t_open + t_tier + t_up, 1e-9), "loader_threads": int(os.environ.get("SS_LOADER_THREADS", 0)) or None,
                   "note": "index_bin_open_s = walk of levels / segments / key heads + sort by key; decode_and_upload_s = postings and positions "
                           "of every key decoded on the host cores (terms in parallel) + image build + sparse-tier append; postings_per_s over "
                           "open + tier + decode_and_upload"}
    say("open", out["open"])
    # ---- queries drawn from the docs themselves (so that they have answers)
    tid_of = {}

    def tid(key):  # term id(s) of a corpus key in the image: (ids, idfs or None)
        if key not in tid_of:
            tid_of[key] = ix.terms_of_key(T.key_hash(key))
        return tid_of[key]
    ands, ors, phrases = [], [], []
    gone_set = set(int(x) for x in gone)
    while len(ands) < n_queries or len(ors) < n_queries or len(phrases) < n_queries:
        d = int(rng.integers(0, n_docs))
        toks = T.doc_tokens(d)
        if len(toks) < 12 or d in gone_set:  # (a tombstoned doc may be a query's only match)
            continue
        mid = [int(r) for r in dict.fromkeys(toks.tolist()) if 20 <= r < 5000]
        if len(ands) < n_queries and len(mid) >= 2:
            ands.append([mid[0], mid[1]])
        uniq = [int(r) for r in dict.fromkeys(toks.tolist())]
        if len(ors) < n_queries and len(uniq) >= 6:
            pick = [uniq[i] for i in rng.choice(len(uniq), 3, replace=False)]  # any ranks: rare terms live in the sparse tier
            ors.append(pick)
        if len(phrases) < n_queries:
            ptoks = toks
            if MF:  # a phrase stands inside ONE field
                ptoks = T.doc_field_tokens(d, int(rng.integers(0, n_fields)))
                if len(ptoks) < 5:
                    ptoks = T.doc_field_tokens(d, 1)
            st = int(rng.integers(0, len(ptoks) - 4))
            ln = int(rng.integers(2, 5))
            win = [int(r) for r in ptoks[st:st + ln]]
            ents = T.query_entries(win)
            ok = len(ents) >= 2 and all(e[0] is not None and T.key_df(e[0]) > 0 for e in ents)  # keys of either tier (a rare word: the sparse phrase kernel)
            if ok and sum(len(e[1]) for e in ents) <= N.SS_MAX_PHRASE:
                phrases.append(ents)
    idf_of = {}

    def single(r):
        return tid(r)[0][0]

    def entry(e):
        comp = tid(e[0])
        idf_of.update({t: i for t, i in comp if i is not None})
        return tuple(t for t, _ in comp) if len(comp) > 1 else comp[0][0]
    q_and = sh.make_queries([[single(r) for r in q] for q in ands], S.QueryType.Intersection)
    q_or = sh.make_queries([[single(r) for r in q] for q in ors], S.QueryType.Union)
    q_ph = sh.make_queries([[entry(e) for e in q] for q in phrases], S.QueryType.Phrase, idf_of=idf_of)
    qv = O.vec_gen(O.VECQ_SEED, 0, n_queries, dim)
    res = {}
    lat = {}
    legs = [("and2", q_and), ("or3", q_or), ("phrase", q_ph)]
    if MF:  # every term must stand in the body (field 1): intersections under a field filter, sparse-tier terms included
        legs.append(("and2_body", sh.make_queries([[single(r) for r in q] for q in ands], S.QueryType.Intersection, field_filter=[1])))
    for name, q in legs:
        res[name] = sh.search_lexical_batch(q, k, S.ResultType.TopkCount)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < seconds / 2:
            sh.search_lexical_batch(q, k, S.ResultType.TopkCount)
            reps += 1
        dt_ = time.perf_counter() - t0
        ab_ = sh.algorithmic_bytes(q, res[name][3], k, n_docs)  # SURVEY 8d bytes of one call, from the exact counts
        lat[name] = {"value": reps * len(q) / dt_, "unit": "queries/s", "queries_per_call": len(q),
                     "entry_point": "ss_bm25_search (host pointers, host clock), TopkCount",
                     "roofline": {"bound": "hbm", "achieved": ab_ * reps / dt_ / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": ab_ * reps / dt_ / 1e9 / 8000.0,
                                  "algorithmic_bytes_per_call": ab_, "ms_per_call": dt_ / reps * 1e3,
                                  "clock": "host clock around the whole call (copies in / out, every kernel of the tiered search): at "
                                           f"{len(q)} queries of a {n_docs}-doc shard a call is launch latency, not bandwidth -- the fraction says how far"}}
    if probe:  # where a small call's time goes: call sizes x result types, tiered and all-dense queries apart (host clock, Python caller)
        pr = {}
        for name, q, qs in (("and2", q_and, ands), ("or3", q_or, ors), ("phrase", q_ph, None)):
            sp = np.array([any(int(t) >= n_dense for t in q["term"][i][:int(q["n_terms"][i])]) for i in range(len(q))])
            for label, sel in (("all", np.arange(len(q))), ("tiered", np.nonzero(sp)[0]), ("dense", np.nonzero(~sp)[0])):
                for nb in (1, 8, 32, 64):
                    if len(sel) < nb:
                        continue
                    for rt in (S.ResultType.Topk, S.ResultType.TopkCount):
                        if isinstance(probe, str) and probe not in f"{name}/{label}/{nb}/{rt.name}":
                            continue
                        qq = q[sel[:nb]]
                        sh.search_lexical_batch(qq, k, rt, reference_shortcuts=False)
                        b0 = sh.one_launch_batches()
                        sh.profile(True)
                        sh.profile_read(0)
                        t0 = time.perf_counter()
                        for _ in range(200):
                            sh.search_lexical_batch(qq, k, rt, reference_shortcuts=False)
                        dt_ = time.perf_counter() - t0
                        pn, pms = sh.profile_read(0)
                        sh.profile(False)
                        pr[f"{name}/{label}/{nb}/{rt.name}"] = {"us_per_call": dt_ / 200 * 1e6, "one_launch": sh.one_launch_batches() - b0,
                                                               "kernel_us": pms / max(pn, 1) * 1e3, "kernels": pn}
        out["probe"] = pr
        for kk_, v in pr.items():
            say(kk_, round(v["us_per_call"], 1), v["one_launch"], "kernel us", round(v["kernel_us"], 1), v["kernels"])
    res["vec"] = sh.search_vector_batch(qv, k)
    # hybrid: the OR query + the vector, RRF over the two top-k lists (search.rs:1962-2035)
    hyb = [S.merge_results(S.SearchMode.Hybrid, (res["or3"][0][i][:res["or3"][2][i]].astype(np.uint64), res["or3"][1][i][:res["or3"][2][i]]),
                           (res["vec"][0][i][:res["vec"][2][i]].astype(np.uint64), res["vec"][1][i][:res["vec"][2][i]]), 0, k) for i in range(n_queries)]
    if MF:
        lat["mean_and_body_matches"] = float(res["and2_body"][3].mean())
    out["queries"] = dict(lat, n=n_queries, mean_and_matches=float(res["and2"][3].mean()), mean_or_matches=float(res["or3"][3].mean()),
                          mean_phrase_matches=float(res["phrase"][3].mean()),
                          phrases_with_ngram_keys=int(sum(any(len(e[1]) > 1 for e in q) for q in phrases)),
                          phrases_naming_a_sparse_term=int(sum(any(tid(e[0])[0][0] >= n_dense for e in q) for q in phrases)),
                          ors_naming_a_sparse_term=int(sum(any(single(r) >= n_dense for r in q) for q in ors)))
    say("queries", out["queries"])
    # ---- parity: every query against the oracle on the corpus' own lists
    if parity and MF:
        t0 = time.perf_counter()
        keys = sorted({r for q in ands + ors for r in q} | {e[0] for q in phrases for e in q})
        o_offs, o_docs, o_flds, o_tfs, o_cnt, o_pos, o_id = [0], [], [], [], [], [], {}
        for key in keys:
            for c in range(len(tid(key))):
                docs, flds, tfs, cnt, pos = T.key_entries(key, c, positions=(c == 0))
                o_id[(key, c)] = len(o_offs) - 1
                o_docs.append(docs); o_flds.append(flds); o_tfs.append(tfs); o_cnt.append(cnt); o_pos.append(pos)
                o_offs.append(o_offs[-1] + len(docs))
        A = dict(offs=np.asarray(o_offs, np.uint64), docs=np.concatenate(o_docs), flds=np.concatenate(o_flds), tfs=np.concatenate(o_tfs),
                 cnt=np.concatenate(o_cnt), pos=np.concatenate(o_pos))
        dlf = T.doclen_fields
        gone_l = [int(x) for x in gone]
        for name, qs, oop, filt in (("and2", ands, O.OP_AND, ()), ("or3", ors, O.OP_OR, ()), ("and2_body", ands, O.OP_AND, (1,))):
            doc, score, cnt, tot = res[name]
            for i, q in enumerate(qs):
                od, os_, otot, _ = O.search_fields_exhaustive(n_docs, dlf, boost, A["offs"], A["docs"], A["flds"], A["tfs"], [o_id[(r, 0)] for r in q], oop,
                                                              k, (), gone_l, field_filter=filt)
                assert int(tot[i]) == otot, f"real format ({n_fields} fields), {name} query {i}: count {int(tot[i])} vs oracle {otot}"
                F.check_topk(doc[i][:cnt[i]], score[i][:cnt[i]], od, os_, 1e-4, f"real format ({n_fields} fields), {name} query {i}")
        doc, score, cnt, tot = res["phrase"]
        for i, q in enumerate(phrases):
            uniq, seq, places, idf, at = [], [], [], [], 0
            for e in q:
                comp = tid(e[0])
                lists = [o_id[(e[0], c)] for c in range(len(comp))]
                for c, l in enumerate(lists):
                    if l not in uniq:
                        uniq.append(l)
                        a_, b_ = int(A["offs"][l]), int(A["offs"][l + 1])
                        idf.append(comp[c][1] if comp[c][1] is not None else float(idf_f32(n_docs, len(np.unique(A["docs"][a_:b_])))))
                seq.append(uniq.index(lists[0])); places.append(at)
                at += len(e[1])
            od, os_, otot = O.search_fields_phrase_items(n_docs, dlf, boost, A["offs"], A["docs"], A["flds"], A["tfs"], A["cnt"], A["pos"], uniq, seq, places,
                                                         k, idf=idf, deleted=gone_l, reference_loop=False)
            assert otot >= 1 and int(tot[i]) == otot, f"real format ({n_fields} fields), phrase {i}: count {int(tot[i])} vs oracle {otot}"
            F.check_topk(doc[i][:cnt[i]], score[i][:cnt[i]], od, os_, 1e-4, f"real format ({n_fields} fields), phrase {i}")
        vd, vs, vc, _ = res["vec"]
        for i in range(min(n_queries, 16)):
            od, os_, _, _ = O.vec_search(rows, qv[i], k, deleted=gone)
            F.check_topk(vd[i][:vc[i]], vs[i][:vc[i]], od, os_, 1e-4, f"real format, vector query {i}")
            ol = (res["or3"][0][i][:res["or3"][2][i]].astype(np.uint64), res["or3"][1][i][:res["or3"][2][i]])
            hd, hs, _ = O.merge(2, ol, (vd[i][:vc[i]].astype(np.uint64), vs[i][:vc[i]]), 0, k)
            assert np.array_equal(hyb[i][0], hd) and np.allclose(hyb[i][1], hs, rtol=1e-6), f"real format, hybrid query {i}"
        out["parity"] = {"queries": {"and2": len(ands), "or3": len(ors), "and2_body": len(ands), "phrase": len(phrases), "vector": min(n_queries, 16),
                                     "hybrid": min(n_queries, 16)},
                         "checked": "exact result_count_total, top-k ids outside the tie band, scores rtol 1e-4 against the BM25F oracle over the mini "
                                    "indexer's own (doc, field) entries, positions and n-gram component field vectors; phrases inside one field; "
                                    "intersections under a field filter; 0.5 % of the docs tombstoned through delete.bin", "seconds": time.perf_counter() - t0}
        say("parity", out["parity"])
    elif parity:
        t0 = time.perf_counter()
        keys = sorted({r for q in ands + ors for r in q} | {e[0] for q in phrases for e in q})
        o_offs, o_docs, o_tfs, o_pos, o_cnt, o_id = [0], [], [], [], [], {}
        for key in keys:
            nc = len(tid(key))
            for c in range(nc):
                docs, tfs, cnt, pos = T.key_postings(key, c, positions=(c == 0))
                o_id[(key, c)] = len(o_offs) - 1
                o_docs.append(docs); o_tfs.append(tfs)
                o_cnt.append(cnt if c == 0 else np.zeros(len(docs), np.uint16))
                if c == 0:
                    o_pos.append(pos)
                o_offs.append(o_offs[-1] + len(docs))
        osh = O.Shard(n_docs, T.doclen, np.asarray(o_offs, np.uint64), np.concatenate(o_docs), np.concatenate(o_tfs))
        osh.set_positions(np.concatenate(o_pos), np.concatenate(o_cnt))
        osh.set_deleted(gone)
        for name, qs, oop in (("and2", ands, O.OP_AND), ("or3", ors, O.OP_OR)):
            doc, score, cnt, tot = res[name]
            for i, q in enumerate(qs):
                od, os_, otot = osh.search_exhaustive([o_id[(r, 0)] for r in q], oop, k)
                assert int(tot[i]) == otot, f"real format, {name} query {i}: count {int(tot[i])} vs oracle {otot}"
                F.check_topk(doc[i][:cnt[i]], score[i][:cnt[i]], od, os_, 1e-4, f"real format, {name} query {i}")
        doc, score, cnt, tot = res["phrase"]
        for i, q in enumerate(phrases):
            uniq, seq, places, idf, at = [], [], [], [], 0
            for e in q:
                comp = tid(e[0])
                lists = [o_id[(e[0], c)] for c in range(len(comp))]
                for c, l in enumerate(lists):
                    if l not in uniq:
                        uniq.append(l)
                        idf.append(comp[c][1] if comp[c][1] is not None else float(idf_f32(n_docs, osh.df(l))))
                seq.append(uniq.index(lists[0])); places.append(at)
                at += len(e[1])
            od, os_, otot = osh.search_phrase_items(uniq, seq, places, k, idf=idf)
            assert otot >= 1 and int(tot[i]) == otot, f"real format, phrase {i}: count {int(tot[i])} vs oracle {otot}"
            F.check_topk(doc[i][:cnt[i]], score[i][:cnt[i]], od, os_, 1e-4, f"real format, phrase {i}")
        vd, vs, vc, _ = res["vec"]
        for i in range(min(n_queries, 16)):
            od, os_, _, _ = O.vec_search(rows, qv[i], k, deleted=gone)
            F.check_topk(vd[i][:vc[i]], vs[i][:vc[i]], od, os_, 1e-4, f"real format, vector query {i}")
            ol = (res["or3"][0][i][:res["or3"][2][i]].astype(np.uint64), res["or3"][1][i][:res["or3"][2][i]])
            hd, hs, _ = O.merge(2, ol, (vd[i][:vc[i]].astype(np.uint64), vs[i][:vc[i]]), 0, k)
            assert np.array_equal(hyb[i][0], hd) and np.allclose(hyb[i][1], hs, rtol=1e-6), f"real format, hybrid query {i}"
        out["parity"] = {"queries": {"and2": len(ands), "or3": len(ors), "phrase": len(phrases), "vector": min(n_queries, 16), "hybrid": min(n_queries, 16)},
                         "checked": "exact result_count_total, top-k ids outside the tie band, scores rtol 1e-4 (lexical); top-k rows / scores (vector); "
                                    "RRF ids and scores (hybrid); oracle lists = the mini indexer's own postings, positions and n-gram component tfs; "
                                    "0.5 % of the docs tombstoned through delete.bin", "seconds": time.perf_counter() - t0}
        say("parity", out["parity"])
    # ---- the reference's calling pattern: `callers` threads, one query per call, Index::search of the C++ mirror
    if callers:
        HL = C.CDLL(os.path.join(ROOT, "seekstorm_amd", "lib", "libseekstorm_host.so"))
        HL.ssh_index_adopt.restype = C.c_void_p
        HL.ssh_index_adopt.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        HL.ssh_index_destroy.argtypes = [C.c_void_p]
        HL.ssh_bench_concurrent.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_double, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
        hs_, hd_ = (C.c_void_p * 1)(sh._h), (C.c_int * 1)(0)
        ixp = HL.ssh_index_adopt(1, hs_, hd_)
        conc = {}
        for name, mode, qs, qt in (("and2", N.MODE_LEXICAL, ands, S.QueryType.Intersection), ("or3", N.MODE_LEXICAL, ors, S.QueryType.Union),
                                   ("hybrid", N.MODE_HYBRID, ors, S.QueryType.Union)):
            flat = np.array([single(r) for q in qs for r in q], np.uint32)
            toff = np.zeros(len(qs) + 1, np.uint32)
            toff[1:] = np.cumsum([len(q) for q in qs])
            o5 = (C.c_double * 5)()
            N.check(HL.ssh_bench_concurrent(ixp, mode, callers, float(seconds), len(qs), flat.ctypes.data, toff.ctypes.data, qv.ctypes.data, int(qt), k,
                                            N.RT_TOPKCOUNT, o5), "ssh_bench_concurrent")
            conc[name] = {"value": o5[0] / o5[1], "unit": "queries/s", "threads": callers, "latency_us_p50": o5[2], "latency_us_p99": o5[3], "errors": int(o5[4])}
            assert o5[4] == 0, f"real format, concurrent {name}: {int(o5[4])} searches failed"
        # ... phrases whose entries are plain keys (an n-gram key arrives at the mirror as its component ids: the batch legs above), and the
        # reference's call shape alone: ONE caller, one query per call -- submit -> results on the host (VERDICT r5 "next" 2: <= 80 us)
        plain = [[e[0] for e in q] for q in phrases if all(len(tid(e[0])) == 1 for e in q)]
        single_caller = {}
        for name, mode, qs, qt, thr in (("phrase", N.MODE_LEXICAL, plain, S.QueryType.Phrase, callers), ("and2", N.MODE_LEXICAL, ands, S.QueryType.Intersection, 1),
                                        ("or3", N.MODE_LEXICAL, ors, S.QueryType.Union, 1), ("phrase", N.MODE_LEXICAL, plain, S.QueryType.Phrase, 1)):
            if not qs:
                continue
            flat = np.array([single(r) for q in qs for r in q], np.uint32)
            toff = np.zeros(len(qs) + 1, np.uint32)
            toff[1:] = np.cumsum([len(q) for q in qs])
            o5 = (C.c_double * 5)()
            before = sh.one_launch_batches()
            N.check(HL.ssh_bench_concurrent(ixp, mode, thr, float(seconds), len(qs), flat.ctypes.data, toff.ctypes.data, qv.ctypes.data, int(qt), k,
                                            N.RT_TOPKCOUNT, o5), "ssh_bench_concurrent")
            rec = {"value": o5[0] / o5[1], "unit": "queries/s", "threads": thr, "latency_us_p50": o5[2], "latency_us_p99": o5[3], "errors": int(o5[4]),
                   "queries": len(qs), "one_launch_batches": sh.one_launch_batches() - before}
            assert o5[4] == 0, f"real format, {thr} caller(s) {name}: {int(o5[4])} searches failed"
            (conc if thr != 1 else single_caller)[name] = rec
        HL.ssh_index_destroy(ixp)
        out["single_caller"] = single_caller
        say("single caller", single_caller)
        out["concurrent_callers"] = conc
        say("concurrent", conc)
    sh.close()
    ix.close()
    T.close()
    return out


if __name__ == "__main__":
    import json
    import sys
    sys.path.insert(0, ROOT)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    nf = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    r = run(n_docs=n, vocab=n, n_fields=nf, log=lambda *a: print(*a, flush=True), probe=(sys.argv[sys.argv.index("--probe") + 1] if "--probe" in sys.argv and len(sys.argv) > sys.argv.index("--probe") + 1 else "--probe" in sys.argv), callers=0 if "--probe" in sys.argv else 64,
            parity="--probe" not in sys.argv)
    print(json.dumps(r))
