"""Full-size oracle answers for the BASELINE configs (C2: 10 M docs, C3: 10 M x 768, C4: hybrid).

TEST INFRASTRUCTURE ONLY (like everything under oracle/): used by tests/test_gpu_fullsize.py and by bench.py as the
CHECKER of the full-size runs -- never as the thing measured.

The corpora are the device generators' own streams regenerated on the host (so_lex_* / so_vec_gen are bit-identical to
ss_bm25_synth / ss_vec_synth), restricted to what a sample of queries needs:
  * C2: only the posting lists of the sample's terms are generated (the other 4 000 lists cannot change an answer);
  * C3: the rows are streamed in slices, every slice scanned by so_vec_search (TopK::push, vector.rs:410-496) and the
    running top-k kept across slices in row order -- the reference's sequential scan, cut into pieces.
A shard of a partitioned corpus (doc g -> shard g % S, local id g // S, index.rs:5284) is (shard_id, n_shards).
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import oracle as O


def host_threads(cap=64):
    return max(1, min(os.cpu_count() or 1, cap))


def effective_cpus():
    """the CPUs this process can really use: min(affinity mask, cgroup CPU quota) -- os.cpu_count() reports the box's logical
    CPUs, which a container with a CFS quota cannot all run on.  -> (cpus as float, {"cpu_count", "affinity", "quota"})"""
    info = {"cpu_count": os.cpu_count() or 1, "affinity": None, "quota": None}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:  # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            info["quota"] = float(q) / float(per)
    except Exception:
        try:  # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                info["quota"] = q / per
        except Exception:
            pass
    eff = float(min(x for x in (info["cpu_count"], info["affinity"], info["quota"]) if x))
    return eff, info


def c2_corpus(n_docs_shard, terms, thresholds, seed=O.LEX_SEED, part=(0, 1), threads=None):
    """decoded postings of `terms` for shard `part` of the generator stream -> (doclen, offs, docs, tfs), CSR row i <-> terms[i]"""
    sid, S = part
    n_global = n_docs_shard * S  # every shard of the bench holds n_docs_shard docs: global ids [0, n_docs_shard * S)
    threads = threads or host_threads()
    dl = O.lex_doclen(n_global, seed)[sid::S][:n_docs_shard]
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(lambda t: O.lex_term(int(t), thresholds[int(t)], n_global, seed), terms))
    dd, tt, offs = [], [], [0]
    for d, f in parts:
        if S > 1:
            m = (d % S) == sid
            d, f = d[m] // S, f[m]
        dd.append(d.astype(np.uint32))
        tt.append(f)
        offs.append(offs[-1] + len(d))
    docs = np.concatenate(dd) if dd else np.zeros(0, np.uint32)
    tfs = np.concatenate(tt) if tt else np.zeros(0, np.uint16)
    return np.ascontiguousarray(dl), np.asarray(offs, np.uint64), docs, tfs.astype(np.uint16)


def c2_answers(n_docs_shard, term_lists, thresholds, k, op=O.OP_OR, rt=O.RT_TOPKCOUNT, seed=O.LEX_SEED, part=(0, 1), threads=None,
               structured=True, not_lists=None, deleted=None):
    """oracle answers of the sampled queries on the full-size shard: list of (docs, scores, total).
    structured: the reference-structured dispatch (union_docid_3 ...), else the union_scan table scan -- same answers.
    not_lists: per query the NOT terms (not_query_list); deleted: shard-local tombstoned doc ids (delete_hashset)."""
    threads = threads or host_threads()
    not_lists = not_lists if not_lists is not None else [[] for _ in term_lists]
    voc = sorted({int(t) for tl in term_lists for t in tl} | {int(t) for tl in not_lists for t in tl})
    dl, offs, docs, tfs = c2_corpus(n_docs_shard, voc, thresholds, seed, part, threads)
    sh = O.Shard(n_docs_shard, dl, offs, docs, tfs)
    if deleted is not None and len(deleted):
        sh.set_deleted(deleted)
    remap = {t: i for i, t in enumerate(voc)}
    qs = [([remap[int(t)] for t in tl], [remap[int(t)] for t in nl]) for tl, nl in zip(term_lists, not_lists)]
    fn = sh.search_ref if structured else sh.search
    with ThreadPoolExecutor(threads) as ex:
        out = list(ex.map(lambda q: fn(q[0], op, k, rt, not_terms=q[1]), qs))
    return out, sh, remap


def c2_answers_chunked(n_docs_shard, term_lists, thresholds, k, op=O.OP_OR, rt=O.RT_TOPKCOUNT, seed=O.LEX_SEED, part=(0, 1), threads=None,
                       chunk=125):
    """c2_answers for MANY queries: the posting lists of a chunk's terms are generated, its queries answered, the lists dropped --
    1000 C2 queries touch ~2000 of the 4096 lists (1e9 postings), a chunk of 125 about a tenth of that"""
    out = []
    for i in range(0, len(term_lists), chunk):
        ans, _, _ = c2_answers(n_docs_shard, term_lists[i:i + chunk], thresholds, k, op, rt, seed, part, threads)
        out += ans
    return out


def _merge_running(best, new, k):
    """running TopK across slices in row order: (score desc, row asc) -- TopK::push admits only score > minimum, so of two
    equal scores the earlier row stays (vector.rs:423-426)"""
    if best is None:
        return new
    d = np.concatenate([best[0], new[0]])
    s = np.concatenate([best[1], new[1]])
    order = np.lexsort((d, -s.astype(np.float64)))[:k]
    return d[order], s[order]


def c3_answers(n_rows_shard, dim, queries, k, seed=O.VEC_SEED, part=(0, 1), threads=None, slice_rows=131072, i8=False,
               simd_order=True):
    """streamed oracle scan of the full-size matrix: per query (rows, scores) of the exact top-k.
    i8: rows / queries quantised with quantize_f32_to_i8, score = integer dot as f32."""
    sid, S = part
    threads = threads or host_threads()
    queries = np.ascontiguousarray(queries, np.float32)
    q8 = O.quantize_i8(queries) if i8 else None
    nq = len(queries)
    starts = list(range(0, n_rows_shard, slice_rows))

    def do_slice(r0):
        n = min(slice_rows, n_rows_shard - r0)
        if S == 1:
            rows = O.vec_gen(seed, r0, n, dim)
        else:  # local rows r0 .. r0+n are global rows r * S + sid
            rows = O.vec_gen_strided(seed, r0 * S + sid, S, n, dim)
        ids = np.arange(r0, r0 + n, dtype=np.uint32)
        out = []
        if i8:
            r8 = O.quantize_i8(rows)
            for qi in range(nq):
                d, s, _, _ = O.vec_search_i8(r8, q8[qi], k, row_doc_ids=ids)
                out.append((d, s))
        else:
            for qi in range(nq):
                d, s, _, _ = O.vec_search(rows, queries[qi], k, row_doc_ids=ids, simd_order=simd_order)
                out.append((d, s))
        return out

    best = [None] * nq
    with ThreadPoolExecutor(threads) as ex:
        for res in ex.map(do_slice, starts):  # results come back in slice order
            for qi in range(nq):
                best[qi] = _merge_running(best[qi], res[qi], k)
    return best


def check_topk(got_doc, got_score, ref_doc, ref_score, rtol=1e-4, what=""):
    """GPU list vs oracle list: same length, scores within rtol position by position, and identical doc ids outside tie
    bands -- a doc may differ only if its score is within rtol of the k-th score or of its neighbour's (equal scores are
    ordered by the tie rule, which the reference leaves to heap order)."""
    got_doc, ref_doc = np.asarray(got_doc, np.int64), np.asarray(ref_doc, np.int64)
    got_score, ref_score = np.asarray(got_score, np.float64), np.asarray(ref_score, np.float64)
    assert len(got_doc) == len(ref_doc), f"{what}: {len(got_doc)} results, oracle {len(ref_doc)}"
    if not len(ref_doc):
        return
    assert np.allclose(got_score, ref_score, rtol=rtol, atol=1e-6), \
        f"{what}: scores differ, max rel {np.max(np.abs(got_score - ref_score) / np.maximum(np.abs(ref_score), 1e-9)):.3g}"
    kth = ref_score[-1]
    band = rtol * max(abs(kth), 1e-9) * 2
    sure_ref = set(ref_doc[ref_score > kth + band].tolist())
    sure_got = set(got_doc[got_score > kth + band].tolist())
    assert sure_ref <= set(got_doc.tolist()), f"{what}: oracle docs above the tie band missing: {sorted(sure_ref - set(got_doc.tolist()))[:5]}"
    assert sure_got <= set(ref_doc.tolist()), f"{what}: docs above the tie band the oracle does not have: {sorted(sure_got - set(ref_doc.tolist()))[:5]}"
