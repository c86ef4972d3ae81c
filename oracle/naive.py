"""Second, deliberately naive restatement (numpy float32) of the scoring rules.

TEST INFRASTRUCTURE ONLY.  It shares no code with ss_oracle.c: two independent restatements
agreeing is our substitute for the Rust binary that cannot be built here.
Citations are file:line under /root/reference/seekstorm/src.
"""
import numpy as np

K = np.float32(1.2)  # add_result.rs:20
B = np.float32(0.75)  # add_result.rs:21


def int_to_byte4(i):  # index.rs:4237-4251
    if i < 24:
        return i
    ii = i - 24
    nb = ii.bit_length()
    if nb < 4:
        return 24 + ii
    sh = nb - 4
    return 24 + (((ii >> sh) & 7) | ((sh + 1) << 3))


def byte4_to_int(b):  # index.rs:4255-4268
    if b < 24:
        return b
    i = b - 24
    bits, sh = i & 7, i >> 3
    return 24 + bits if sh == 0 else 24 + ((bits | 8) << (sh - 1))


DLC = np.array([byte4_to_int(b) for b in range(256)], np.uint32)  # index.rs:4271-4279


def component_cache(avgdl):  # commit.rs:321-325
    q = DLC.astype(np.float32) / np.float32(avgdl)
    return (K * (np.float32(1.0) - B + B * q)).astype(np.float32)


def avgdl(doclen_bytes):  # commit.rs:318-319
    s = int(DLC[doclen_bytes].astype(np.uint64).sum())
    return np.float32(s) / np.float32(len(doclen_bytes))


def idf(N, n):  # search.rs:3225-3230
    Nf, nf = np.float32(N), np.float32(n)
    return np.log(((Nf - nf + np.float32(0.5)) / (nf + np.float32(0.5))) + np.float32(1.0), dtype=np.float32)


def bm25_scores(n_docs, doclen_bytes, postings, op_and):
    """postings: list of (docs u32, tfs) per query term.  Returns (doc ids, scores) of all matches,
    scores summed in query-term order in float32 (add_result.rs:1435-1449)."""
    comp = component_cache(avgdl(doclen_bytes))
    sc = np.zeros(n_docs, np.float32)
    cnt = np.zeros(n_docs, np.int32)
    for docs, tfs in postings:
        i = idf(n_docs, len(docs))
        tf = tfs.astype(np.float32)
        c = comp[doclen_bytes[docs]]
        w = (i * ((tf * (K + np.float32(1.0))) / (tf + c))).astype(np.float32)
        sc[docs] = sc[docs] + w
        cnt[docs] += 1
    m = (cnt == len(postings)) if op_and else (cnt > 0)
    ids = np.nonzero(m)[0].astype(np.uint32)
    return ids, sc[ids]


def topk(ids, scores, k):
    """exact top-k by (score desc, id asc)"""
    order = np.lexsort((ids, -scores.astype(np.float64)))[:k]
    return ids[order], scores[order]


def cosine_topk(rows, q, k):
    s = (rows.astype(np.float32) @ q.astype(np.float32)).astype(np.float32)
    ids = np.arange(len(s), dtype=np.uint32)
    return topk(ids, s, k)


def rrf(lex_ids, vec_ids, length):  # search.rs:1962-2035 (k = 0.6, 0-based ranks)
    sc = {}
    for i, d in enumerate(lex_ids):
        sc[int(d)] = np.float32(1.0) / (np.float32(0.6) + np.float32(i))
    for i, d in enumerate(vec_ids):
        r = np.float32(1.0) / (np.float32(0.6) + np.float32(i))
        sc[int(d)] = np.float32(sc[int(d)] + r) if int(d) in sc else r
    items = sorted(sc.items(), key=lambda kv: (-float(kv[1]), kv[0]))[:length]
    return [d for d, _ in items], [s for _, s in items]
