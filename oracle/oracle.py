"""ctypes front-end of the CPU oracle (oracle/ss_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (seekstorm_amd) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

OP_AND, OP_OR = 0, 1
RT_COUNT, RT_TOPK, RT_TOPKCOUNT = 0, 1, 2
CT_ARRAY, CT_BITMAP, CT_RLE = 1, 2, 3

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        build()
    L = C.CDLL(_LIB)
    L.so_int_to_byte4.restype = C.c_uint8
    L.so_int_to_byte4.argtypes = [C.c_uint32]
    L.so_byte4_to_int.restype = C.c_uint32
    L.so_byte4_to_int.argtypes = [C.c_uint8]
    L.so_avgdl.restype = C.c_float
    L.so_avgdl.argtypes = [C.c_uint64, C.c_uint64]
    L.so_bm25_component_cache.argtypes = [C.c_float, f32p]
    L.so_idf.restype = C.c_float
    L.so_idf.argtypes = [C.c_uint64, C.c_uint64]
    L.so_bm25_term.restype = C.c_float
    L.so_bm25_term.argtypes = [C.c_float, C.c_uint32, C.c_float]
    L.so_splitmix64.restype = C.c_uint64
    L.so_splitmix64.argtypes = [C.c_uint64]
    L.so_h.restype = C.c_uint64
    L.so_h.argtypes = [C.c_uint64] * 3
    L.so_lex_doclen.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, u8p, u8p]
    L.so_lex_term_postings.restype = C.c_uint64
    L.so_lex_term_postings.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, u32p, u16p, C.c_uint64]
    L.so_vec_gen.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, f32p]
    L.so_vec_gen_strided.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, f32p]
    L.so_vec_gen_strided.restype = None
    L.so_geom06.argtypes = [C.c_uint32]
    L.so_geom06.restype = C.c_uint32
    L.so_shard_build.restype = C.c_void_p
    L.so_shard_build.argtypes = [C.c_uint64, u8p, C.c_uint32, u64p, u32p, u16p]
    L.so_shard_free.argtypes = [C.c_void_p]
    L.so_shard_avgdl.restype = C.c_float
    L.so_shard_avgdl.argtypes = [C.c_void_p]
    L.so_shard_posting_count.restype = C.c_uint64
    L.so_shard_posting_count.argtypes = [C.c_void_p, C.c_uint32]
    L.so_shard_container.restype = C.c_int
    L.so_shard_container.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, u32p, u32p, f32p]
    L.so_shard_decode_block.restype = C.c_uint32
    L.so_shard_decode_block.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, u16p]
    L.so_search_lex_not.argtypes = [C.c_void_p, C.c_uint32, u32p, C.c_uint32, u32p, C.c_int, C.c_uint32, C.c_int, u32p, f32p, u64p]
    L.so_search_lex_exhaustive_not.argtypes = [C.c_void_p, C.c_uint32, u32p, C.c_uint32, u32p, C.c_int, C.c_uint32, u32p, f32p, u64p]
    for fn in (L.so_search_lex, L.so_search_lex_exhaustive, L.so_search_lex_not, L.so_search_lex_exhaustive_not):
        fn.restype = C.c_uint32
    L.so_search_lex.argtypes = [C.c_void_p, C.c_uint32, u32p, C.c_int, C.c_uint32, C.c_int, u32p, f32p, u64p]
    L.so_search_lex_exhaustive.argtypes = [C.c_void_p, C.c_uint32, u32p, C.c_int, C.c_uint32, u32p, f32p, u64p]
    L.so_query_stats.argtypes = [C.c_void_p, C.c_uint32, u32p, u64p, u64p]
    L.so_normalize_f32.argtypes = [f32p, C.c_uint32]
    L.so_dot_f32.restype = C.c_float
    L.so_dot_f32.argtypes = [f32p, f32p, C.c_uint32]
    L.so_dot_f32_lanes8.restype = C.c_float
    L.so_dot_f32_lanes8.argtypes = [f32p, f32p, C.c_uint32]
    L.so_vec_search.restype = C.c_uint32
    L.so_vec_search.argtypes = [f32p, C.c_uint64, C.c_uint32, u32p, f32p, C.c_uint32, C.c_float, C.c_int,
                                u32p, f32p, u64p, u64p]
    L.so_vec_search_del.restype = C.c_uint32
    L.so_vec_search_del.argtypes = [f32p, C.c_uint64, C.c_uint32, u32p, f32p, C.c_uint32, C.c_float, C.c_int, u64p,
                                    C.c_uint64, u32p, f32p, u64p, u64p]
    L.so_search_fields_exhaustive.restype = C.c_uint32
    L.so_search_fields_exhaustive.argtypes = [C.c_uint64, C.c_uint32, u8p, f32p, u64p, u32p, u8p, u16p, C.c_uint32, u32p,
                                              C.c_uint32, u32p, C.c_int, C.c_uint32, u64p, C.c_uint64, u32p, f32p, u64p, f32p]
    L.so_quantize_f32_to_i8.argtypes = [f32p, C.c_uint32, C.c_void_p]
    L.so_vec_search_i8.restype = C.c_uint32
    L.so_vec_search_i8.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, u32p, f32p, C.c_void_p, C.c_int, C.c_float, C.c_uint32,
                                   C.c_float, u64p, C.c_uint64, u32p, f32p, u64p, u64p]
    L.so_shard_set_deleted.argtypes = [C.c_void_p, u64p, C.c_uint64]
    L.so_vector_score_field.restype = C.c_float
    L.so_vector_score_field.argtypes = [C.c_float]
    L.so_threshold_raw.restype = C.c_float
    L.so_threshold_raw.argtypes = [C.c_float]
    L.so_merge.restype = C.c_uint32
    L.so_merge.argtypes = [C.c_int, u64p, f32p, C.c_uint32, u64p, f32p, C.c_uint32, C.c_uint32, C.c_uint32,
                           u64p, f32p, u8p]
    _lib = L
    return L


# ---------------------------------------------------------------- synthetic corpora
LEX_SEED = 0x5EEC5701
LEX_SEED_CLUSTERED = LEX_SEED | (1 << 63)  # bit 63: a term's density varies with the doc's cluster (so_lex_cluster_thresh)
VEC_SEED = 0xC051AE01
VECQ_SEED = 0xC051AE02
N_VOCAB = 4096


def len_table():
    """1024 log-normal quantiles of doc length, clamp [8,2000], as SmallFloat bytes."""
    from statistics import NormalDist
    nd = NormalDist()
    L = lib()
    out = np.zeros(1024, np.uint8)
    for i in range(1024):
        z = nd.inv_cdf((i + 0.5) / 1024.0)
        ln = int(round(np.exp(np.log(120.0) + 0.6 * z)))
        ln = min(max(ln, 8), 2000)
        out[i] = L.so_int_to_byte4(ln)
    return out


def term_thresholds(n_vocab=N_VOCAB):
    """df_t/N log-uniform in [0.05%, 20%] by term index, as 32-bit compare thresholds."""
    t = np.arange(n_vocab, dtype=np.float64)
    frac = 0.0005 * (0.2 / 0.0005) ** (t / max(n_vocab - 1, 1))
    return np.minimum(np.floor(frac * 2.0 ** 32), 2.0 ** 32 - 1).astype(np.uint32)


def lex_doclen(n_docs, seed=LEX_SEED):
    tab = len_table()
    out = np.empty(n_docs, np.uint8)
    lib().so_lex_doclen(seed, 0, n_docs, _p(tab, u8p), _p(out, u8p))
    return out


def lex_term(term, thresh32, n_docs, seed=LEX_SEED):
    L = lib()
    n = L.so_lex_term_postings(seed, term, int(thresh32), n_docs, None, None, 0)
    docs = np.empty(n, np.uint32)
    tfs = np.empty(n, np.uint16)
    L.so_lex_term_postings(seed, term, int(thresh32), n_docs, _p(docs, u32p), _p(tfs, u16p), n)
    return docs, tfs


def lex_corpus(n_docs, terms, seed=LEX_SEED, thresholds=None):
    """Decoded postings (CSR) of the given vocabulary term ids; CSR row i <-> terms[i]."""
    th = term_thresholds() if thresholds is None else thresholds
    offs = [0]
    dl, tl = [], []
    for t in terms:
        d, f = lex_term(int(t), th[int(t)], n_docs, seed)
        dl.append(d)
        tl.append(f)
        offs.append(offs[-1] + len(d))
    docs = np.concatenate(dl) if dl else np.zeros(0, np.uint32)
    tfs = np.concatenate(tl) if tl else np.zeros(0, np.uint16)
    return np.asarray(offs, np.uint64), docs.astype(np.uint32), tfs.astype(np.uint16)


def vec_gen(seed, r0, n, dim, normalize=True):
    out = np.empty((n, dim), np.float32)
    lib().so_vec_gen(seed, r0, n, dim, 1 if normalize else 0, _p(out, f32p))
    return out


def vec_gen_strided(seed, r0, stride, n, dim, normalize=True):
    """rows r0, r0 + stride, ...: one shard's rows of a partitioned generator stream"""
    out = np.empty((n, dim), np.float32)
    lib().so_vec_gen_strided(seed, r0, stride, n, dim, 1 if normalize else 0, _p(out, f32p))
    return out


# ---------------------------------------------------------------- shard wrapper
class Shard:
    def __init__(self, n_docs, doclen, offs, docs, tfs):
        self.n_docs = int(n_docs)
        self.doclen = np.ascontiguousarray(doclen, np.uint8)
        self.offs = np.ascontiguousarray(offs, np.uint64)
        self.docs = np.ascontiguousarray(docs, np.uint32)
        self.tfs = np.ascontiguousarray(tfs, np.uint16)
        self.n_terms = len(self.offs) - 1
        self.h = lib().so_shard_build(self.n_docs, _p(self.doclen, u8p), self.n_terms, _p(self.offs, u64p),
                                      _p(self.docs, u32p), _p(self.tfs, u16p))

    def set_deleted(self, doc_ids):
        ids = np.ascontiguousarray(doc_ids, np.uint64)
        lib().so_shard_set_deleted(self.h, _p(ids, u64p) if len(ids) else None, len(ids))

    def __del__(self):
        try:
            lib().so_shard_free(self.h)
        except Exception:
            pass

    @property
    def avgdl(self):
        return lib().so_shard_avgdl(self.h)

    def df(self, t):
        return lib().so_shard_posting_count(self.h, t)

    def container(self, t, bo):
        bid, cnt, mp = C.c_uint32(), C.c_uint32(), C.c_float()
        ct = lib().so_shard_container(self.h, t, bo, C.byref(bid), C.byref(cnt), C.byref(mp))
        return ct, bid.value, cnt.value, mp.value

    def decode_block(self, t, bo):
        out = np.empty(65536, np.uint16)
        n = lib().so_shard_decode_block(self.h, t, bo, _p(out, u16p))
        return out[:n].copy()

    def _search(self, fn, terms, not_terms, op, k, rt=None):
        q = np.ascontiguousarray(terms, np.uint32)
        nq = np.ascontiguousarray(not_terms, np.uint32)
        od = np.empty(max(k, 1), np.uint32)
        os_ = np.empty(max(k, 1), np.float32)
        tot = C.c_uint64()
        head = (self.h, len(q), _p(q, u32p), len(nq), _p(nq, u32p) if len(nq) else None, op, k)
        if rt is None:
            n = fn(*head, _p(od, u32p), _p(os_, f32p), C.byref(tot))
        else:
            n = fn(*head, rt, _p(od, u32p), _p(os_, f32p), C.byref(tot))
        return od[:n].copy(), os_[:n].copy(), tot.value

    def search(self, terms, op, k, rt=RT_TOPKCOUNT, not_terms=()):
        return self._search(lib().so_search_lex_not, terms, not_terms, op, k, rt)

    def search_ref(self, terms, op, k, rt=RT_TOPKCOUNT, not_terms=()):
        """the dispatch as the reference structures it (single_blockid / union_docid_2 / union_docid_3 / ...): the function
        the CPU baseline times; must agree with search()"""
        f = lib().so_search_lex_ref
        f.restype = C.c_uint32
        f.argtypes = [C.c_void_p, C.c_uint32, u32p, C.c_uint32, u32p, C.c_int, C.c_uint32, C.c_int, u32p, f32p, u64p]
        return self._search(f, terms, not_terms, op, k, rt)

    def set_positions(self, positions, counts=None):
        """positions of every posting in CSR order; counts: positions per posting where that is not the tf (the component lists
        of an n-gram key: the key's positions behind its first component, 0 for the others)"""
        ps = np.ascontiguousarray(positions, np.uint16)
        f = lib().so_shard_set_positions_counts
        f.restype = None
        f.argtypes = [C.c_void_p, u16p, C.c_uint64, u16p]
        cn = None if counts is None else np.ascontiguousarray(counts, np.uint16)
        f(self.h, _p(ps, u16p), len(ps), None if cn is None else _p(cn, u16p))

    def search_phrase_items(self, terms, seq, places, k, idf=None, reference_loop=True):
        """QueryType::Phrase with n-gram keys among its entries: terms = unique terms (an n-gram key: its component lists), seq[i] =
        index into terms of entry i (an n-gram key: its first component), places[i] = term_index_nonunique of entry i, idf = per
        unique term or None -> (docs, scores, matches)"""
        q = np.ascontiguousarray(terms, np.uint32)
        sq = np.ascontiguousarray(seq, np.uint8)
        pl = np.ascontiguousarray(places, np.uint8)
        assert len(pl) == len(sq)
        idf_a = None if idf is None else np.ascontiguousarray(idf, np.float32)
        od = np.empty(max(k, 1), np.uint32)
        os_ = np.empty(max(k, 1), np.float32)
        tot = C.c_uint64()
        f = lib().so_search_phrase_items
        f.restype = C.c_uint32
        f.argtypes = [C.c_void_p, C.c_uint32, u32p, f32p, C.c_uint32, u8p, u8p, C.c_uint32, C.c_int, u32p, f32p, C.POINTER(C.c_uint64)]
        n = f(self.h, len(q), _p(q, u32p), None if idf_a is None else _p(idf_a, f32p), len(sq), _p(sq, u8p), _p(pl, u8p), k,
              1 if reference_loop else 0, _p(od, u32p), _p(os_, f32p), C.byref(tot))
        return od[:n].copy(), os_[:n].copy(), tot.value

    def search_phrase(self, terms, seq, k, reference_loop=True):
        """QueryType::Phrase: terms = unique terms, seq = index into terms of every word -> (docs, scores, matches)"""
        q = np.ascontiguousarray(terms, np.uint32)
        sq = np.ascontiguousarray(seq, np.uint8)
        od = np.empty(max(k, 1), np.uint32)
        os_ = np.empty(max(k, 1), np.float32)
        tot = C.c_uint64()
        f = lib().so_search_phrase
        f.restype = C.c_uint32
        f.argtypes = [C.c_void_p, C.c_uint32, u32p, C.c_uint32, u8p, C.c_uint32, C.c_int, u32p, f32p, C.POINTER(C.c_uint64)]
        n = f(self.h, len(q), _p(q, u32p), len(sq), _p(sq, u8p), k, 1 if reference_loop else 0, _p(od, u32p), _p(os_, f32p), C.byref(tot))
        return od[:n].copy(), os_[:n].copy(), tot.value

    def all_terms_frequent(self, terms, top_k):
        f = lib().so_all_terms_frequent
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_uint32, u32p, C.c_uint32]
        q = np.ascontiguousarray(terms, np.uint32)
        return bool(f(self.h, len(q), _p(q, u32p), int(top_k)))

    def search_exhaustive(self, terms, op, k, not_terms=(), idf=None, reference_shortcuts=False):
        """idf: the idf of every term, given (n-gram component terms carry idf_ngram_i); None = from the lists' own counts.
        reference_shortcuts: apply all_terms_frequent when its condition holds (intersection.rs:198-209)"""
        if reference_shortcuts:
            q = np.ascontiguousarray(terms, np.uint32)
            nq = np.ascontiguousarray(not_terms, np.uint32)
            w = None if idf is None else np.ascontiguousarray(idf, np.float32)
            od = np.empty(max(k, 1), np.uint32)
            os_ = np.empty(max(k, 1), np.float32)
            tot = C.c_uint64()
            f = lib().so_search_lex_exhaustive_opt
            f.restype = C.c_uint32
            f.argtypes = [C.c_void_p, C.c_uint32, u32p, f32p, C.c_uint32, u32p, C.c_int, C.c_uint32, C.c_int, u32p, f32p,
                          C.POINTER(C.c_uint64)]
            sc = 1 if (op == OP_AND and len(q) > 1 and self.all_terms_frequent(q, k)) else 0
            n = f(self.h, len(q), _p(q, u32p), _p(w, f32p), len(nq), _p(nq, u32p) if len(nq) else None, op, k, sc, _p(od, u32p),
                  _p(os_, f32p), C.byref(tot))
            return od[:n].copy(), os_[:n].copy(), tot.value
        if idf is None:
            return self._search(lib().so_search_lex_exhaustive_not, terms, not_terms, op, k)
        q = np.ascontiguousarray(terms, np.uint32)
        w = np.ascontiguousarray(idf, np.float32)
        assert len(w) == len(q)
        nq = np.ascontiguousarray(not_terms, np.uint32)
        od = np.empty(max(k, 1), np.uint32)
        os_ = np.empty(max(k, 1), np.float32)
        tot = C.c_uint64()
        f = lib().so_search_lex_exhaustive_idf
        f.restype = C.c_uint32
        f.argtypes = [C.c_void_p, C.c_uint32, u32p, f32p, C.c_uint32, u32p, C.c_int, C.c_uint32, u32p, f32p, C.POINTER(C.c_uint64)]
        n = f(self.h, len(q), _p(q, u32p), _p(w, f32p), len(nq), _p(nq, u32p) if len(nq) else None, op, k, _p(od, u32p),
              _p(os_, f32p), C.byref(tot))
        return od[:n].copy(), os_[:n].copy(), tot.value

    def stats(self, terms):
        q = np.ascontiguousarray(terms, np.uint32)
        a, b = C.c_uint64(), C.c_uint64()
        lib().so_query_stats(self.h, len(q), _p(q, u32p), C.byref(a), C.byref(b))
        return a.value, b.value


def split_corpus(n_docs, doclen, offs, docs, tfs, n_shards):
    """the reference's document partitioning: doc g -> shard g % S with local id g // S (index.rs:5284); per shard
    (n_docs, doclen, offs, docs, tfs)"""
    n_terms = len(offs) - 1
    S = int(n_shards)
    term_of = np.repeat(np.arange(n_terms, dtype=np.int64), np.diff(offs.astype(np.int64)))
    sh_of = (docs % S).astype(np.int64)
    order = np.argsort(sh_of, kind="stable")  # shard-major; inside a shard the (term, doc) order is kept
    cnt = np.bincount(sh_of * n_terms + term_of, minlength=S * n_terms).reshape(S, n_terms)
    starts = np.concatenate([[0], np.cumsum(cnt.sum(axis=1))])
    d_sorted = (docs[order] // S).astype(np.uint32)
    t_sorted = tfs[order]
    out = []
    for sh in range(S):
        o = np.zeros(n_terms + 1, np.uint64)
        o[1:] = np.cumsum(cnt[sh])
        a, b = int(starts[sh]), int(starts[sh + 1])
        out.append(((n_docs - sh + S - 1) // S, np.ascontiguousarray(doclen[sh::S]), o, d_sorted[a:b], t_sorted[a:b]))
    return out


def bench_lex(shards, queries, op, k, rt, mode, threads, seconds):
    """CPU baseline harness (so_bench_lex): shards = list of Shard, queries = [nq][nt] term ids.
    -> (queries/s, queries answered, latencies in microseconds (latency mode))"""
    L = lib()
    f = L.so_bench_lex
    f.restype = C.c_double
    f.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, u32p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int,
                  C.c_uint32, C.c_double, u64p, C.POINTER(C.c_double), C.c_uint32, u32p]
    hs = (C.c_void_p * len(shards))(*[sh.h for sh in shards])
    q = np.ascontiguousarray(queries, np.uint32)
    assert q.ndim == 2
    lat = np.zeros(1 << 20, np.float64)
    done, nlat = C.c_uint64(), C.c_uint32()
    qps = f(hs, len(shards), _p(q.reshape(-1), u32p), q.shape[0], q.shape[1], op, k, rt, mode, threads, float(seconds),
            C.byref(done), lat.ctypes.data_as(C.POINTER(C.c_double)), len(lat), C.byref(nlat))
    return qps, done.value, lat[:nlat.value].copy()


def vec_search(rows, query, k, row_doc_ids=None, threshold_raw=-3.4028234663852886e38, simd_order=True, deleted=None):
    rows = np.ascontiguousarray(rows, np.float32)
    query = np.ascontiguousarray(query, np.float32)
    rd = None if row_doc_ids is None else np.ascontiguousarray(row_doc_ids, np.uint32)
    od = np.empty(max(k, 1), np.uint32)
    os_ = np.empty(max(k, 1), np.float32)
    tot, obs = C.c_uint64(), C.c_uint64()
    dl = np.unique(np.ascontiguousarray([] if deleted is None else deleted, np.uint64))  # sorted
    n = lib().so_vec_search_del(_p(rows, f32p), rows.shape[0], rows.shape[1], _p(rd, u32p), _p(query, f32p), k,
                                threshold_raw, 1 if simd_order else 0, _p(dl, u64p) if len(dl) else None, len(dl),
                                _p(od, u32p), _p(os_, f32p), C.byref(tot), C.byref(obs))
    return od[:n].copy(), os_[:n].copy(), tot.value, obs.value


def search_fields_exhaustive(n_docs, doclen_fields, boost, offs, docs, fields, tfs, terms, op, k, not_terms=(), deleted=(),
                             field_filter=()):
    """BM25F ground truth over several indexed fields (add_result.rs:1171-1426) -> (doc ids, scores, total, avgdl)"""
    dl = np.ascontiguousarray(doclen_fields, np.uint8)
    b = None if boost is None else np.ascontiguousarray(boost, np.float32)
    offs = np.ascontiguousarray(offs, np.uint64)
    docs = np.ascontiguousarray(docs, np.uint32)
    fields = np.ascontiguousarray(fields, np.uint8)
    tfs = np.ascontiguousarray(tfs, np.uint16)
    q = np.ascontiguousarray(terms, np.uint32)
    nq_ = np.ascontiguousarray(not_terms, np.uint32)
    de = np.ascontiguousarray(deleted, np.uint64)
    od = np.empty(max(k, 1), np.uint32)
    os_ = np.empty(max(k, 1), np.float32)
    tot, avg = C.c_uint64(), C.c_float()
    mask = 0
    for f_ in field_filter:
        mask |= 1 << int(f_)
    assert not mask or op == OP_AND or len(q) == 1, "field filter: intersections and single-term queries"
    fn = lib().so_search_fields_filtered
    fn.restype = C.c_uint32
    fn.argtypes = [C.c_uint64, C.c_uint32, u8p, f32p, u64p, u32p, u8p, u16p, C.c_uint32, u32p, C.c_uint32, u32p, C.c_int,
                   C.c_uint32, u64p, C.c_uint64, C.c_uint32, u32p, f32p, C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
    n = fn(n_docs, dl.shape[0], _p(dl.reshape(-1), u8p), _p(b, f32p), _p(offs, u64p), _p(docs, u32p),
           _p(fields, u8p), _p(tfs, u16p), len(q), _p(q, u32p), len(nq_), _p(nq_, u32p) if len(nq_) else None, op, k,
           _p(de, u64p) if len(de) else None, len(de), mask, _p(od, u32p), _p(os_, f32p), C.byref(tot), C.byref(avg))
    return od[:n].copy(), os_[:n].copy(), tot.value, avg.value


def search_fields_shortcut(n_docs, doclen_fields, boost, offs, docs, fields, tfs, terms, k, deleted=()):
    """intersection under all_terms_frequent, several indexed fields (add_result.rs:1595-1607) -> (doc ids, scores, total)"""
    dl = np.ascontiguousarray(doclen_fields, np.uint8)
    b = None if boost is None else np.ascontiguousarray(boost, np.float32)
    offs = np.ascontiguousarray(offs, np.uint64)
    docs = np.ascontiguousarray(docs, np.uint32)
    fields = np.ascontiguousarray(fields, np.uint8)
    tfs = np.ascontiguousarray(tfs, np.uint16)
    q = np.ascontiguousarray(terms, np.uint32)
    de = np.ascontiguousarray(deleted, np.uint64)
    od = np.empty(max(k, 1), np.uint32)
    os_ = np.empty(max(k, 1), np.float32)
    tot = C.c_uint64()
    fn = lib().so_search_fields_shortcut
    fn.restype = C.c_uint32
    fn.argtypes = [C.c_uint64, C.c_uint32, u8p, f32p, u64p, u32p, u8p, u16p, C.c_uint32, u32p, C.c_uint32, u64p, C.c_uint64, u32p, f32p,
                   C.POINTER(C.c_uint64)]
    n = fn(n_docs, dl.shape[0], _p(dl.reshape(-1), u8p), _p(b, f32p), _p(offs, u64p), _p(docs, u32p), _p(fields, u8p), _p(tfs, u16p),
           len(q), _p(q, u32p), k, _p(de, u64p) if len(de) else None, len(de), _p(od, u32p), _p(os_, f32p), C.byref(tot))
    return od[:n].copy(), os_[:n].copy(), tot.value


def search_fields_phrase(n_docs, doclen_fields, boost, offs, docs, fields, tfs, positions, terms, seq, k, deleted=(), field_filter=(),
                         reference_loop=True):
    """phrase over several indexed fields (add_result.rs:2964-3414) -> (doc ids, scores, total)
    positions: flat u16 array, for every (term, doc, field) entry in CSR order its tf positions inside the field"""
    dl = np.ascontiguousarray(doclen_fields, np.uint8)
    b = None if boost is None else np.ascontiguousarray(boost, np.float32)
    offs = np.ascontiguousarray(offs, np.uint64)
    docs = np.ascontiguousarray(docs, np.uint32)
    fields = np.ascontiguousarray(fields, np.uint8)
    tfs = np.ascontiguousarray(tfs, np.uint16)
    pos = np.ascontiguousarray(positions, np.uint16)
    assert len(pos) == int(tfs.astype(np.uint64).sum())
    q = np.ascontiguousarray(terms, np.uint32)
    sq = np.ascontiguousarray(seq, np.uint8)
    de = np.ascontiguousarray(deleted, np.uint64)
    od = np.empty(max(k, 1), np.uint32)
    os_ = np.empty(max(k, 1), np.float32)
    tot = C.c_uint64()
    mask = 0
    for f_ in field_filter:
        mask |= 1 << int(f_)
    fn = lib().so_search_fields_phrase
    fn.restype = C.c_uint32
    fn.argtypes = [C.c_uint64, C.c_uint32, u8p, f32p, u64p, u32p, u8p, u16p, u16p, C.c_uint32, u32p, C.c_uint32, u8p, C.c_uint32,
                   u64p, C.c_uint64, C.c_uint32, C.c_int, u32p, f32p, C.POINTER(C.c_uint64)]
    n = fn(n_docs, dl.shape[0], _p(dl.reshape(-1), u8p), _p(b, f32p), _p(offs, u64p), _p(docs, u32p), _p(fields, u8p), _p(tfs, u16p),
           _p(pos, u16p), len(q), _p(q, u32p), len(sq), _p(sq, u8p), k, _p(de, u64p) if len(de) else None, len(de), mask,
           1 if reference_loop else 0, _p(od, u32p), _p(os_, f32p), C.byref(tot))
    return od[:n].copy(), os_[:n].copy(), tot.value


def search_fields_phrase_items(n_docs, doclen_fields, boost, offs, docs, fields, tfs, counts, positions, terms, seq, places, k, idf=None,
                               deleted=(), field_filter=(), reference_loop=True):
    """search_fields_phrase with n-gram keys among the entries: counts = positions behind every (list, doc, field) entry (the key's own
    count with its first component, 0 elsewhere), places = term_index_nonunique of every entry, idf per list (idf_ngram_i)"""
    dl = np.ascontiguousarray(doclen_fields, np.uint8)
    b = None if boost is None else np.ascontiguousarray(boost, np.float32)
    offs = np.ascontiguousarray(offs, np.uint64)
    docs = np.ascontiguousarray(docs, np.uint32)
    fields = np.ascontiguousarray(fields, np.uint8)
    tfs = np.ascontiguousarray(tfs, np.uint16)
    cnt = np.ascontiguousarray(counts, np.uint16)
    pos = np.ascontiguousarray(positions, np.uint16)
    assert len(pos) == int(cnt.astype(np.uint64).sum()) and len(cnt) == len(tfs)
    q = np.ascontiguousarray(terms, np.uint32)
    sq = np.ascontiguousarray(seq, np.uint8)
    pl = np.ascontiguousarray(places, np.uint8)
    idf_a = None if idf is None else np.ascontiguousarray(idf, np.float32)
    de = np.ascontiguousarray(deleted, np.uint64)
    od = np.empty(max(k, 1), np.uint32)
    os_ = np.empty(max(k, 1), np.float32)
    tot = C.c_uint64()
    mask = 0
    for f_ in field_filter:
        mask |= 1 << int(f_)
    fn = lib().so_search_fields_phrase_items
    fn.restype = C.c_uint32
    fn.argtypes = [C.c_uint64, C.c_uint32, u8p, f32p, u64p, u32p, u8p, u16p, u16p, u16p, C.c_uint32, u32p, f32p, C.c_uint32, u8p, u8p, C.c_uint32,
                   u64p, C.c_uint64, C.c_uint32, C.c_int, u32p, f32p, C.POINTER(C.c_uint64)]
    n = fn(n_docs, dl.shape[0], _p(dl.reshape(-1), u8p), _p(b, f32p), _p(offs, u64p), _p(docs, u32p), _p(fields, u8p), _p(tfs, u16p), _p(cnt, u16p),
           _p(pos, u16p), len(q), _p(q, u32p), None if idf_a is None else _p(idf_a, f32p), len(sq), _p(sq, u8p), _p(pl, u8p), k,
           _p(de, u64p) if len(de) else None, len(de), mask, 1 if reference_loop else 0, _p(od, u32p), _p(os_, f32p), C.byref(tot))
    return od[:n].copy(), os_[:n].copy(), tot.value


def quantize_i8(v):
    v = np.ascontiguousarray(v, np.float32)
    out = np.empty(v.shape, np.int8)
    lib().so_quantize_f32_to_i8(_p(v.reshape(-1), f32p), v.size, out.ctypes.data)
    return out


def vec_search_i8(rows_i8, query_i8, k, row_doc_ids=None, row_scale=None, query_scale=None, threshold_raw=-3.4028234663852886e38,
                  deleted=None):
    rows = np.ascontiguousarray(rows_i8, np.int8)
    q = np.ascontiguousarray(query_i8, np.int8)
    rd = None if row_doc_ids is None else np.ascontiguousarray(row_doc_ids, np.uint32)
    rs = None if row_scale is None else np.ascontiguousarray(row_scale, np.float32)
    scaled = row_scale is not None or query_scale is not None
    od = np.empty(max(k, 1), np.uint32)
    os_ = np.empty(max(k, 1), np.float32)
    tot, obs = C.c_uint64(), C.c_uint64()
    dl = np.unique(np.ascontiguousarray([] if deleted is None else deleted, np.uint64))
    n = lib().so_vec_search_i8(rows.ctypes.data, rows.shape[0], rows.shape[1], _p(rd, u32p), _p(rs, f32p), q.ctypes.data,
                               1 if scaled else 0, 1.0 if query_scale is None else float(query_scale), k, threshold_raw,
                               _p(dl, u64p) if len(dl) else None, len(dl), _p(od, u32p), _p(os_, f32p), C.byref(tot), C.byref(obs))
    return od[:n].copy(), os_[:n].copy(), tot.value, obs.value


NO_THRESHOLD = -3.4028234663852886e38


def _ann_args(n_rows, level_clusters, child_count):
    if level_clusters is None:  # AnnMode::All
        return np.zeros(0, np.uint32), np.zeros(0, np.uint32)
    lc = np.ascontiguousarray(level_clusters, np.uint32)
    cc = np.ascontiguousarray(child_count, np.uint32)
    assert int(lc.sum()) == len(cc) and int(cc.sum()) == n_rows and np.all(cc > 0)
    return lc, cc


def _field_args(row_field, fields):
    if row_field is None or not fields:
        return None, 0
    mask = 0
    for f in fields:
        mask |= 1 << int(f)
    return np.ascontiguousarray(row_field, np.uint16), mask


u16p = C.POINTER(C.c_uint16)


def vec_search_ann(rows, query, k, level_clusters, child_count, n_probe=0xFFFFFFFF, cluster_threshold_raw=NO_THRESHOLD,
                   row_doc_ids=None, threshold_raw=NO_THRESHOLD, simd_order=True, deleted=None, row_field=None, fields=()):
    """AnnMode::Nprobe / Similaritythreshold / both (vector.rs:1300-1392) -> (docs, scores, total, observed rows, clusters)"""
    rows = np.ascontiguousarray(rows, np.float32)
    query = np.ascontiguousarray(query, np.float32)
    lc, cc = _ann_args(rows.shape[0], level_clusters, child_count)
    rd = None if row_doc_ids is None else np.ascontiguousarray(row_doc_ids, np.uint32)
    od = np.empty(max(k, 1), np.uint32)
    os_ = np.empty(max(k, 1), np.float32)
    tot, obs, ncl = C.c_uint64(), C.c_uint64(), C.c_uint64()
    dl = np.unique(np.ascontiguousarray([] if deleted is None else deleted, np.uint64))
    f = lib().so_vec_search_ann
    f.restype = C.c_uint32
    f.argtypes = [f32p, C.c_uint64, C.c_uint32, u32p, f32p, C.c_uint32, C.c_float, C.c_int, C.c_uint32, u32p, u32p, C.c_uint32,
                  C.c_float, u64p, C.c_uint64, u16p, C.c_uint64, u32p, f32p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                  C.POINTER(C.c_uint64)]
    rf, fm = _field_args(row_field, fields)
    n = f(_p(rows, f32p), rows.shape[0], rows.shape[1], _p(rd, u32p), _p(query, f32p), k, threshold_raw, 1 if simd_order else 0,
          len(lc), _p(lc, u32p) if len(lc) else None, _p(cc, u32p) if len(cc) else None, n_probe, cluster_threshold_raw,
          _p(dl, u64p) if len(dl) else None, len(dl), _p(rf, u16p), fm, _p(od, u32p), _p(os_, f32p), C.byref(tot), C.byref(obs),
          C.byref(ncl))
    return od[:n].copy(), os_[:n].copy(), tot.value, obs.value, ncl.value


def vec_search_i8_ann(rows_i8, query_i8, k, level_clusters, child_count, n_probe=0xFFFFFFFF, cluster_threshold_raw=NO_THRESHOLD,
                      row_doc_ids=None, row_scale=None, query_scale=None, threshold_raw=NO_THRESHOLD, deleted=None,
                      row_field=None, fields=()):
    rows = np.ascontiguousarray(rows_i8, np.int8)
    q = np.ascontiguousarray(query_i8, np.int8)
    lc, cc = _ann_args(rows.shape[0], level_clusters, child_count)
    rd = None if row_doc_ids is None else np.ascontiguousarray(row_doc_ids, np.uint32)
    rs = None if row_scale is None else np.ascontiguousarray(row_scale, np.float32)
    scaled = row_scale is not None or query_scale is not None
    od = np.empty(max(k, 1), np.uint32)
    os_ = np.empty(max(k, 1), np.float32)
    tot, obs, ncl = C.c_uint64(), C.c_uint64(), C.c_uint64()
    dl = np.unique(np.ascontiguousarray([] if deleted is None else deleted, np.uint64))
    f = lib().so_vec_search_i8_ann
    f.restype = C.c_uint32
    f.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, u32p, f32p, C.c_void_p, C.c_int, C.c_float, C.c_uint32, C.c_float,
                  C.c_uint32, u32p, u32p, C.c_uint32, C.c_float, u64p, C.c_uint64, u16p, C.c_uint64, u32p, f32p,
                  C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    rf, fm = _field_args(row_field, fields)
    n = f(rows.ctypes.data, rows.shape[0], rows.shape[1], _p(rd, u32p), _p(rs, f32p), q.ctypes.data, 1 if scaled else 0,
          1.0 if query_scale is None else float(query_scale), k, threshold_raw, len(lc), _p(lc, u32p) if len(lc) else None,
          _p(cc, u32p) if len(cc) else None, n_probe, cluster_threshold_raw, _p(dl, u64p) if len(dl) else None, len(dl),
          _p(rf, u16p), fm, _p(od, u32p), _p(os_, f32p), C.byref(tot), C.byref(obs), C.byref(ncl))
    return od[:n].copy(), os_[:n].copy(), tot.value, obs.value, ncl.value


def normalize(v):
    v = np.ascontiguousarray(v, np.float32).copy()
    lib().so_normalize_f32(_p(v, f32p), v.shape[-1])
    return v


def merge(mode, lex=None, vec=None, offset=0, length=10):
    ld = np.ascontiguousarray(lex[0] if lex else [], np.uint64)
    ls = np.ascontiguousarray(lex[1] if lex else [], np.float32)
    vd = np.ascontiguousarray(vec[0] if vec else [], np.uint64)
    vs = np.ascontiguousarray(vec[1] if vec else [], np.float32)
    od = np.empty(max(length, 1), np.uint64)
    os_ = np.empty(max(length, 1), np.float32)
    src = np.empty(max(length, 1), np.uint8)
    n = lib().so_merge(mode, _p(ld, u64p), _p(ls, f32p), len(ld), _p(vd, u64p), _p(vs, f32p), len(vd), offset,
                       length, _p(od, u64p), _p(os_, f32p), _p(src, u8p))
    return od[:n].copy(), os_[:n].copy(), src[:n].copy()


def vec_search_euclid(rows, query, k, level_clusters=None, child_count=None, n_probe=0xFFFFFFFF, cluster_threshold_raw=NO_THRESHOLD,
                      row_doc_ids=None, threshold_raw=NO_THRESHOLD, simd_order=True, deleted=None, row_field=None, fields=()):
    """VectorSimilarity::Euclidean, f32: score = -euclidean_f32[_avx2] (vector_similarity.rs:912 / 938); AnnMode::All when
    level_clusters is None -> (docs, scores, total, observed rows, clusters)"""
    rows = np.ascontiguousarray(rows, np.float32)
    query = np.ascontiguousarray(query, np.float32)
    lc, cc = _ann_args(rows.shape[0], level_clusters, child_count)
    rd = None if row_doc_ids is None else np.ascontiguousarray(row_doc_ids, np.uint32)
    od = np.empty(max(k, 1), np.uint32)
    os_ = np.empty(max(k, 1), np.float32)
    tot, obs, ncl = C.c_uint64(), C.c_uint64(), C.c_uint64()
    dl = np.unique(np.ascontiguousarray([] if deleted is None else deleted, np.uint64))
    f = lib().so_vec_search_euclid
    f.restype = C.c_uint32
    f.argtypes = [f32p, C.c_uint64, C.c_uint32, u32p, f32p, C.c_uint32, C.c_float, C.c_int, C.c_uint32, u32p, u32p, C.c_uint32,
                  C.c_float, u64p, C.c_uint64, u16p, C.c_uint64, u32p, f32p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                  C.POINTER(C.c_uint64)]
    rf, fm = _field_args(row_field, fields)
    n = f(_p(rows, f32p), rows.shape[0], rows.shape[1], _p(rd, u32p), _p(query, f32p), k, threshold_raw, 1 if simd_order else 0,
          len(lc), _p(lc, u32p) if len(lc) else None, _p(cc, u32p) if len(cc) else None, n_probe, cluster_threshold_raw,
          _p(dl, u64p) if len(dl) else None, len(dl), _p(rf, u16p), fm, _p(od, u32p), _p(os_, f32p), C.byref(tot), C.byref(obs),
          C.byref(ncl))
    return od[:n].copy(), os_[:n].copy(), tot.value, obs.value, ncl.value


def vec_search_i8_euclid(rows_i8, query_i8, k, level_clusters=None, child_count=None, n_probe=0xFFFFFFFF,
                         cluster_threshold_raw=NO_THRESHOLD, row_doc_ids=None, row_scale=None, row_norm=None, query_scale=None,
                         query_norm=None, threshold_raw=NO_THRESHOLD, deleted=None, row_field=None, fields=()):
    """VectorSimilarity::Euclidean, i8: -euclidean_i8 (exact), or with scales / norms -euclidean_i8_quantized
    (vector_similarity.rs:921 / 1721)"""
    rows = np.ascontiguousarray(rows_i8, np.int8)
    q = np.ascontiguousarray(query_i8, np.int8)
    lc, cc = _ann_args(rows.shape[0], level_clusters, child_count)
    rd = None if row_doc_ids is None else np.ascontiguousarray(row_doc_ids, np.uint32)
    rs = None if row_scale is None else np.ascontiguousarray(row_scale, np.float32)
    rn = None if row_norm is None else np.ascontiguousarray(row_norm, np.float32)
    quantized = row_scale is not None or query_scale is not None
    od = np.empty(max(k, 1), np.uint32)
    os_ = np.empty(max(k, 1), np.float32)
    tot, obs, ncl = C.c_uint64(), C.c_uint64(), C.c_uint64()
    dl = np.unique(np.ascontiguousarray([] if deleted is None else deleted, np.uint64))
    f = lib().so_vec_search_i8_euclid
    f.restype = C.c_uint32
    f.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, u32p, f32p, f32p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_uint32, C.c_float,
                  C.c_uint32, u32p, u32p, C.c_uint32, C.c_float, u64p, C.c_uint64, u16p, C.c_uint64, u32p, f32p,
                  C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    rf, fm = _field_args(row_field, fields)
    n = f(rows.ctypes.data, rows.shape[0], rows.shape[1], _p(rd, u32p), _p(rs, f32p), _p(rn, f32p), q.ctypes.data, 1 if quantized else 0,
          1.0 if query_scale is None else float(query_scale), 0.0 if query_norm is None else float(query_norm), k, threshold_raw,
          len(lc), _p(lc, u32p) if len(lc) else None, _p(cc, u32p) if len(cc) else None, n_probe, cluster_threshold_raw,
          _p(dl, u64p) if len(dl) else None, len(dl), _p(rf, u16p), fm, _p(od, u32p), _p(os_, f32p), C.byref(tot), C.byref(obs),
          C.byref(ncl))
    return od[:n].copy(), os_[:n].copy(), tot.value, obs.value, ncl.value


def euclidean_f32(a, b, simd_order=True):
    """euclidean_f32_avx2 lane order (simd_order, dim % 8 == 0) or euclidean_f32 sequential (vector_similarity.rs:938 / 912)"""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    f = lib().so_euclidean_f32_lanes8 if simd_order else lib().so_euclidean_f32
    f.restype = C.c_float
    f.argtypes = [f32p, f32p, C.c_uint32]
    return float(f(_p(a, f32p), _p(b, f32p), len(a)))


def phrase_match(position_lists, reference_loop=True, places=None):
    """so_phrase_match_places: position_lists[i] = ascending positions of the i-th entry of the phrase, places[i] = its place
    (term_index_nonunique; None = 0, 1, 2, ...: a phrase of single terms)"""
    n = len(position_lists)
    arrs = [np.ascontiguousarray(p, np.uint16) for p in position_lists]
    ptrs = (u16p * n)(*[_p(a, u16p) if len(a) else C.cast(None, u16p) for a in arrs])
    cnt = np.array([len(a) for a in arrs], np.uint32)
    pl = None if places is None else np.ascontiguousarray(places, np.uint32)
    f = lib().so_phrase_match_places
    f.restype = C.c_int
    f.argtypes = [C.c_uint32, C.POINTER(u16p), u32p, u32p, C.c_int]
    return bool(f(n, ptrs, _p(cnt, u32p), None if pl is None else _p(pl, u32p), 1 if reference_loop else 0))


def synth_positions(doclen_bytes, docs, tfs, seed=99):
    """positions for every posting: tf distinct ascending positions below the doc's (decoded) length, deterministic"""
    L = lib()
    dec = np.array([L.so_byte4_to_int(i) for i in range(256)], np.int64)
    rng = np.random.default_rng(seed)
    out = []
    for d, tf in zip(docs, tfs):
        n = max(int(dec[doclen_bytes[d]]), int(tf), 1)
        out.append(np.sort(rng.choice(min(n, 65535), int(tf), replace=False)).astype(np.uint16))
    return np.concatenate(out) if out else np.zeros(0, np.uint16)


def bench_vec(rows, queries, k, mode, threads, seconds):
    """CPU baseline harness of the vector path (so_bench_vec) -> (queries/s, queries answered, latencies in microseconds)"""
    rows = np.ascontiguousarray(rows, np.float32)
    q = np.ascontiguousarray(queries, np.float32)
    f = lib().so_bench_vec
    f.restype = C.c_double
    f.argtypes = [f32p, C.c_uint64, C.c_uint32, f32p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_double, u64p,
                  C.POINTER(C.c_double), C.c_uint32, u32p]
    lat = np.zeros(1 << 16, np.float64)
    done, nlat = C.c_uint64(), C.c_uint32()
    qps = f(_p(rows, f32p), rows.shape[0], rows.shape[1], _p(q, f32p), q.shape[0], k, mode, threads, float(seconds), C.byref(done),
            lat.ctypes.data_as(C.POINTER(C.c_double)), len(lat), C.byref(nlat))
    return qps, done.value, lat[:nlat.value].copy()


def turboquant_i8(rows, seed_mask, avx2=False):
    """TurboQuant::quantize_f32_i8 of every row -> (i8 [n, dim_pow2], scale [n], norm [n])"""
    rows = np.ascontiguousarray(rows, np.float32)
    mask = np.ascontiguousarray(seed_mask, np.float32)
    dim = len(mask)
    f = lib().so_turboquant_i8
    f.restype = None
    f.argtypes = [f32p, C.c_uint32, f32p, C.c_uint32, C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    q = np.zeros((rows.shape[0], dim), np.int8)
    sc = np.zeros(rows.shape[0], np.float32)
    nm = np.zeros(rows.shape[0], np.float32)
    for i in range(rows.shape[0]):
        s_, n_ = C.c_float(), C.c_float()
        f(rows[i].ctypes.data_as(f32p), rows.shape[1], _p(mask, f32p), dim, 1 if avx2 else 0, q[i].ctypes.data, C.byref(s_), C.byref(n_))
        sc[i], nm[i] = s_.value, n_.value
    return q, sc, nm


# ---- Point (geo) facets: geo_search.rs
def morton_encode(lat, lon):
    """encode_morton_2_d of every (lat, lon) -> uint64 codes"""
    f = lib().so_morton_encode
    f.restype = C.c_uint64
    f.argtypes = [C.c_double, C.c_double]
    la, lo = np.atleast_1d(np.asarray(lat, np.float64)), np.atleast_1d(np.asarray(lon, np.float64))
    return np.array([f(float(a), float(b)) for a, b in zip(la, lo)], np.uint64)


def morton_decode(codes):
    f = lib().so_morton_decode
    f.restype = None
    f.argtypes = [C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    out = np.zeros((len(codes), 2), np.float64)
    for i, c in enumerate(codes):
        a, b = C.c_double(), C.c_double()
        f(int(c), C.byref(a), C.byref(b))
        out[i] = a.value, b.value
    return out


def geo_distances(codes, base, unit):
    """unit "km" / "miles": euclidian_distance(base, doc); "sortkey": simplified_distance(doc, base)"""
    f = lib().so_geo_distances
    f.restype = None
    f.argtypes = [C.c_uint64, u64p, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double)]
    c = np.ascontiguousarray(codes, np.uint64)
    out = np.zeros(len(c), np.float64)
    f(len(c), _p(c, u64p), float(base[0]), float(base[1]), {"sortkey": 0, "km": 1, "miles": 2}[unit], out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def geo_morton_range(base, distance, unit):
    f = lib().so_geo_morton_range
    f.restype = None
    f.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, u64p]
    out = np.zeros(2, np.uint64)
    f(float(base[0]), float(base[1]), float(distance), {"km": 1, "miles": 2}[unit], _p(out, u64p))
    return int(out[0]), int(out[1])
