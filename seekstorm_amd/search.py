"""Host-side mirror of the reference's search interface for the query hot path.

Names, argument meaning and error behaviour follow the `seekstorm` crate (citations relative to
/root/reference/seekstorm/src):

  QueryType / ResultType / SearchMode / ResultSource / Result / ResultObject   search.rs:59,168,73,186; min_heap.rs:17-40
  Shard.search_lexical_shard     <- SearchLexicalShard::search_lexical_shard     search.rs:2427-2442
  Shard.search_vector_shard      <- SearchVectorShard::search_vector_shard       vector.rs:1105-1115
  Index.search                   <- <IndexArc as Search>::search                 search.rs:1134-1150, 1154-2131

Out of scope by SURVEY.md section 8 (stays in the Rust host): tokenizer / term hashing, so a query is a list of
resolved term ids instead of a query string; facets, filters, highlights, query rewriting.

All compute goes through the C ABI (seekstorm_amd/_native.py -> libseekstorm_hip.so).  Like the reference's search
path, a failing shard degrades to an empty ResultObject (search.rs:2461-2463, vector.rs:1222-1224) unless
`strict=True` is passed, in which case the C-ABI error is raised.  ONE code is no failure: SS_ENOTSUP says "this query is
the host's own dispatch's to answer" (search.rs:3374-3560 stands in the same function as the seam; INTEGRATION.md section 4
lists the shapes) -- the mirrors have no CPU path by design (the product never computes on the host), so they report it
as ResultObject.cpu_dispatch = True instead of an empty answer that would look like "no hits".
"""
import ctypes as C
import enum
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _native as N


class QueryType(enum.IntEnum):  # search.rs:59
    Union = N.OP_UNION
    Intersection = N.OP_INTERSECTION
    Phrase = N.OP_PHRASE  # "...": the terms of the query are the words of the phrase, in order (a word may repeat)


class ResultType(enum.IntEnum):  # search.rs:168
    Count = N.RT_COUNT
    Topk = N.RT_TOPK
    TopkCount = N.RT_TOPKCOUNT


class SearchMode(enum.IntEnum):  # search.rs:73
    Lexical = N.MODE_LEXICAL
    Vector = N.MODE_VECTOR
    Hybrid = N.MODE_HYBRID


class ResultSource(enum.IntEnum):  # min_heap.rs
    Lexical = N.SRC_LEXICAL
    Vector = N.SRC_VECTOR
    Hybrid = N.SRC_HYBRID


@dataclass
class Result:  # min_heap.rs:17-40 (fields of the hot path)
    doc_id: int
    score: float
    source: ResultSource = ResultSource.Lexical


@dataclass
class ResultObject:  # search.rs:186-213
    results: List[Result] = field(default_factory=list)
    result_count: int = 0
    result_count_total: int = 0
    observed_vector_count: int = 0
    observed_cluster_count: int = 0
    cpu_dispatch: bool = False  # SS_ENOTSUP: not an empty answer -- the reference's own dispatch block (search.rs:3374-3560) answers it


SIMILARITY_NORMALIZATION_64_I8 = np.float32(1.0) / np.float32(16129.0)  # vector.rs:29


def normalize_f32(v):
    """vector_similarity.rs:70-74 (query-side normalisation, search.rs:1464-1475)."""
    v = np.ascontiguousarray(v, np.float32)
    s = np.float32(0.0)
    for x in v:  # sequential f32 sum like the reference
        s = np.float32(s + np.float32(x * x))
    return (v * (np.float32(1.0) / np.sqrt(s, dtype=np.float32))).astype(np.float32)


def quantize_f32_to_i8(v):
    """vector_similarity.rs:1226-1232: (v * 127).round().clamp(-127, 127) as i8 -- f32::round is half away from zero"""
    x = np.ascontiguousarray(v, np.float32) * np.float32(127.0)
    r = np.sign(x) * np.floor(np.abs(x) + np.float32(0.5))
    return np.clip(r, -127, 127).astype(np.int8)


def turboquant_dim(n):
    """TurboQuant::next_power_of_two (vector_similarity.rs:1836-1842): the padded dimension of a TurboQuantI8 index"""
    d = 1
    while d < int(n):
        d <<= 1
    return d


def turboquant_f32_to_i8(v, seed_mask, avx2=False):
    """TurboQuant::quantize_f32_i8 / quantize_f32_i8_avx2 (vector_similarity.rs:1927-1983), the query side of a
    Quantization::TurboQuantI8 index (search.rs:1545-1594): pad to len(seed_mask) = the next power of two, multiply by the +-1
    seed mask, Fast Walsh-Hadamard transform, scale = max(sqrt(sum x^2) / sqrt(dim) / 32, 1e-8), round(x / scale) -> i8.
    -> (i8 vector, scale, norm = sum q^2 * scale^2), to be searched with ss_vec_search_i8 (query_scale) or, under Euclidean,
    ss_vec_search_i8_euclid (query_scale, query_norm).  seed_mask: the index's TurboQuant::seed_mask (ChaCha8Rng::seed_from_u64
    (1234) in the reference -- rand_chacha stays on the host side of the boundary).  avx2: the summation order, reciprocal
    multiplication and packs saturation of the reference's x86 path, which is the one an x86 host indexes with."""
    f = np.float32
    mask = np.ascontiguousarray(seed_mask, f)
    dim = len(mask)
    a = np.zeros(dim, f)
    v = np.ascontiguousarray(v, f)
    a[:min(len(v), dim)] = v[:dim]
    a = (a * mask).astype(f)
    h = 1
    while h < dim:  # butterflies of one stage are independent: elementwise f32 adds, the reference's values bit for bit
        b = a.reshape(dim // (2 * h), 2, h)
        a = np.concatenate([(b[:, 0, :] + b[:, 1, :])[:, None, :], (b[:, 0, :] - b[:, 1, :])[:, None, :]], axis=1).reshape(dim).astype(f)
        h *= 2
    a = (a / np.sqrt(f(dim), dtype=f)).astype(f)
    sq = (a * a).astype(f)
    if avx2 and dim >= 8:
        lanes = np.add.accumulate(sq.reshape(dim // 8, 8), axis=0, dtype=f)[-1]  # eight sequential f32 chains
        x = (lanes[4:] + lanes[:4]).astype(f)
        y = (x[:2] + x[2:]).astype(f)
        sum_sq = f(y[0] + y[1])
    else:
        sum_sq = np.add.accumulate(sq, dtype=f)[-1]  # sequential f32 sum (np.sum adds pairwise)
    scale = f(f(np.sqrt(sum_sq, dtype=f) / np.sqrt(f(dim), dtype=f)) / f(32.0))
    if not scale > f(1e-8):
        scale = f(1e-8)
    if avx2 and dim >= 16:
        s_ = (a * f(f(1.0) / scale)).astype(f)
        adj = (s_ + np.where(np.signbit(s_), f(-0.5), f(0.5)).astype(f)).astype(f)
        q = np.clip(np.trunc(adj), -128, 127).astype(np.int8)
    else:
        x = (a / scale).astype(f)
        r = np.sign(x) * np.floor(np.abs(x) + f(0.5))
        q = np.clip(r, -127, 127).astype(np.int8)
    sqn = int((q.astype(np.int64) ** 2).sum())
    return q, float(scale), float(f(f(f(sqn) * scale) * scale))


def idf_f32(indexed_doc_count, posting_count):
    """search.rs:3225-3230, all f32."""
    Nf, nf = np.float32(indexed_doc_count), np.float32(posting_count)
    return np.float32(np.log(((Nf - nf + np.float32(0.5)) / (nf + np.float32(0.5))) + np.float32(1.0), dtype=np.float32))


def threshold_raw(similarity_threshold, euclidean=False):
    """TopK::new, vector.rs:388-397: Dot / Cosine ((t * 2) - 1) / SIMILARITY_NORMALIZATION_64_I8, Euclidean -t"""
    if similarity_threshold is None:
        return N.FLT_MIN_NEG
    if euclidean:
        return float(-np.float32(similarity_threshold))
    return float(((np.float32(similarity_threshold) * np.float32(2.0)) - np.float32(1.0)) / SIMILARITY_NORMALIZATION_64_I8)


@dataclass(frozen=True)
class AnnMode:
    """search.rs AnnMode: All | Similaritythreshold(t) | Nprobe(n) | NprobeSimilaritythreshold(n, t) (vector.rs:1300-1307).
    n_probe = clusters visited per level (0 = no limit), similarity_threshold = normalised medoid similarity below which a
    cluster is skipped (None = no threshold).  AnnMode.All is None wherever an ann_mode is accepted."""
    n_probe: int = 0
    similarity_threshold: Optional[float] = None
    All = None

    @staticmethod
    def Nprobe(n_probe):
        return AnnMode(int(n_probe), None)

    @staticmethod
    def Similaritythreshold(threshold):
        return AnnMode(0, float(threshold))

    @staticmethod
    def NprobeSimilaritythreshold(n_probe, threshold):
        return AnnMode(int(n_probe), float(threshold))

    def _c(self, euclidean=False):
        if self.n_probe < 0 or (self.n_probe == 0 and self.similarity_threshold is None):
            raise ValueError("AnnMode needs n_probe >= 1 or a similarity threshold")
        return N.AnnModeC(self.n_probe, threshold_raw(self.similarity_threshold, euclidean), 0)


def _vector_options(ann_mode, field_filter, euclidean=False, observed=False):
    """ss_ann_mode of a call: the AnnMode (None = All) and the field filter (indexed field ids, empty = every field);
    observed: also report observed_vector_count (SS_ANN_REPORT_OBSERVED: three words per query in out_clusters)"""
    mask = 0
    for f in field_filter or ():
        if not 0 <= int(f) < 64:
            raise ValueError("field filter: indexed field ids 0..63")
        mask |= 1 << int(f)
    if ann_mode is None and not mask and not observed:
        return None
    m = N.AnnModeC(0, N.FLT_MIN_NEG, 0) if ann_mode is None else ann_mode._c(euclidean)
    m.field_mask = mask
    m.flags = N.SS_ANN_REPORT_OBSERVED if observed else 0
    return m


def _split_clusters(ncl, nq, observed):
    """out_clusters of a call -> (observed_cluster_count [nq], observed_vector_count [nq] or None)"""
    if not observed:
        return ncl, None
    t = ncl.reshape(nq, 3)
    return t[:, 0].copy(), t[:, 1].astype(np.uint64) | (t[:, 2].astype(np.uint64) << np.uint64(32))


class IndexBin:
    """Parsed view of a shard's index.bin (ss_index_bin_*; host only -- works without a GPU).  Term id = rank of the
    key_hash among the keys; `term_of_key` is the lookup the Rust side does after hashing the term.  An n-gram key
    (key_hash & 7 != 0) holds one term id per component term, consecutive: `terms_of_key` returns them with the idf each
    is scored with (idf_ngram_i, search.rs:3231-3262)."""

    def __init__(self, data, indexed_field_count=1, key_head_size=20, segment_number_bits=11, min_posting_count=0):
        self._buf = np.frombuffer(bytes(data), np.uint8).copy()  # the handle borrows these bytes
        self.indexed_field_count = int(indexed_field_count)
        h = C.c_void_p()
        N.check(N.lib().ss_index_bin_open(self._buf.ctypes.data, len(self._buf), indexed_field_count, key_head_size,
                                          segment_number_bits, C.byref(h)), "ss_index_bin_open")
        self._h = h
        if min_posting_count:  # device image for the frequent terms only; the tail stays with the host's own path
            N.check(N.lib().ss_index_bin_filter(h, int(min_posting_count), None), "ss_index_bin_filter")
        nd, ps = C.c_uint64(), C.c_uint64()
        nl, nt, ng = C.c_uint32(), C.c_uint32(), C.c_uint32()
        N.check(N.lib().ss_index_bin_info(h, C.byref(nd), C.byref(ps), C.byref(nl), C.byref(nt), C.byref(ng)), "ss_index_bin_info")
        self.indexed_doc_count, self.positions_sum_normalized = nd.value, ps.value
        self.level_count, self.term_count, self.ngram_keys_skipped = nl.value, nt.value, ng.value
        self.term_keys = np.zeros(self.term_count, np.uint64)
        if self.term_count:
            N.check(N.lib().ss_index_bin_term_keys(h, N.ptr(self.term_keys, N.u64p)), "ss_index_bin_term_keys")
        self.term_components = np.ones(self.term_count, np.uint8)   # components of the term's key (1 = SingleTerm)
        self.term_component = np.zeros(self.term_count, np.uint8)   # which of them this term is
        self.term_component_df = np.zeros(self.term_count, np.uint32)  # posting count of the component term (n-gram keys)
        if self.term_count:
            N.check(N.lib().ss_index_bin_term_ngram(h, N.ptr(self.term_components, N.u8p), N.ptr(self.term_component, N.u8p),
                                                    N.ptr(self.term_component_df, N.u32p)), "ss_index_bin_term_ngram")

    def tier(self, dense_min_posting_count):
        """keys with fewer postings go to the image's SPARSE tier instead of being dropped (ss_index_bin_tier): term ids = the frequent
        keys in hash order, then the rare keys in hash order; returns the first sparse term id"""
        nd = C.c_uint32()
        N.check(N.lib().ss_index_bin_tier(self._h, int(dense_min_posting_count), C.byref(nd)), "ss_index_bin_tier")
        self.n_dense = int(nd.value)
        N.check(N.lib().ss_index_bin_term_keys(self._h, N.ptr(self.term_keys, N.u64p)), "ss_index_bin_term_keys")
        N.check(N.lib().ss_index_bin_term_ngram(self._h, N.ptr(self.term_components, N.u8p), N.ptr(self.term_component, N.u8p),
                                                N.ptr(self.term_component_df, N.u32p)), "ss_index_bin_term_ngram")
        return self.n_dense

    def term_of_key(self, key_hash):
        """term id of a key (first component for an n-gram key), None if the image does not hold it; a tiered index (tier())
        keeps its keys sorted inside each tier: one binary search per tier"""
        nd = getattr(self, "n_dense", None)
        for lo, hi in (((0, self.term_count),) if nd is None else ((0, nd), (nd, self.term_count))):
            i = lo + int(np.searchsorted(self.term_keys[lo:hi], np.uint64(key_hash)))
            if i < hi and int(self.term_keys[i]) == int(key_hash):
                return i
        return None

    def terms_of_key(self, key_hash):
        """[(term id, idf)] a query term with this key contributes: one entry with idf None (= from the list's own posting
        count, as for any term) for a SingleTerm key, one per component with idf_ngram_i for an n-gram key"""
        t = self.term_of_key(key_hash)
        if t is None:
            return None
        n = int(self.term_components[t])
        if n == 1:
            return [(t, None)]
        return [(t + c, float(idf_f32(self.indexed_doc_count, int(self.term_component_df[t + c])))) for c in range(n)]

    def postings(self, term):
        n = C.c_uint64()
        N.lib().ss_index_bin_term_postings(self._h, term, 0, None, None, C.byref(n))
        docs, tfs = np.zeros(n.value, np.uint32), np.zeros(n.value, np.uint16)
        N.check(N.lib().ss_index_bin_term_postings(self._h, term, n.value, N.ptr(docs, N.u32p), N.ptr(tfs, N.u16p), C.byref(n)),
                "ss_index_bin_term_postings")
        return docs, tfs

    def decode_all(self, positions=False):
        """every key's postings, decoded on the loader's worker threads (one indexed field): (offs, docs, tfs) or, with positions,
        (offs, docs, tfs, npos, positions)"""
        a, b = C.c_uint64(), C.c_uint64()
        N.check(N.lib().ss_index_bin_decode_stats(self._h, 1 if positions else 0, C.byref(a), C.byref(b)), "ss_index_bin_decode_stats")
        offs = np.zeros(self.term_count + 1, np.uint64)
        docs, tfs = np.zeros(a.value, np.uint32), np.zeros(a.value, np.uint16)
        npos = np.zeros(a.value if positions else 0, np.uint16)
        pos = np.zeros(max(b.value, 1) if positions else 0, np.uint16)
        N.check(N.lib().ss_index_bin_decode_all(self._h, N.ptr(offs, N.u64p), N.ptr(docs, N.u32p), N.ptr(tfs, N.u16p), a.value,
                                                N.ptr(npos, N.u16p) if positions else None, N.ptr(pos, N.u16p) if positions else None,
                                                b.value), "ss_index_bin_decode_all")
        return (offs, docs, tfs, npos, pos[:b.value]) if positions else (offs, docs, tfs)

    def close(self):
        if self._h:
            N.lib().ss_index_bin_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Shard:
    """One shard image on one MI355X (opaque ss_shard handle)."""
    compose_filtered_unions = False  # True: unions under a field filter through 2^n - 1 filtered intersections (rounds 1-2) instead of the kernels

    def __init__(self, device=0, shard_id=0):
        self.device = device
        self.shard_id = shard_id
        h = C.c_void_p()
        N.check(N.lib().ss_shard_create(device, C.byref(h)), "ss_shard_create")
        self._h = h
        self._df_cache = {}
        self.indexed_doc_count = 0
        self.lexical_field_count = 1
        self.vector_count = 0
        self.dim = 0
        self.vector_precision = "f32"  # "i8": Precision::I8 image, queries are quantised with quantize_f32_to_i8
        self.vector_euclidean = False  # set_vector_similarity("euclidean")

    def close(self):
        if getattr(self, "_h", None):
            N.lib().ss_shard_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- image (re)build: end of open_shard (index.rs:3796) / after commit (commit.rs:142-148)
    def upload_lexical(self, n_docs, doclen_bytes, term_offsets, doc_ids, tfs, positions=None):
        """positions: for every posting in CSR order its tf positions (ascending) -- needed by phrase queries only"""
        dl = np.ascontiguousarray(doclen_bytes, np.uint8)
        off = np.ascontiguousarray(term_offsets, np.uint64)
        d = np.ascontiguousarray(doc_ids, np.uint32)
        t = np.ascontiguousarray(tfs, np.uint16)
        if positions is None:
            N.check(N.lib().ss_bm25_upload(self._h, int(n_docs), N.ptr(dl, N.u8p), len(off) - 1, N.ptr(off, N.u64p),
                                           N.ptr(d, N.u32p), N.ptr(t, N.u16p)), "ss_bm25_upload")
        else:
            ps = np.ascontiguousarray(positions, np.uint16)
            N.check(N.lib().ss_bm25_upload_positions(self._h, int(n_docs), N.ptr(dl, N.u8p), len(off) - 1, N.ptr(off, N.u64p),
                                                     N.ptr(d, N.u32p), N.ptr(t, N.u16p), N.ptr(ps, N.u16p), len(ps)),
                    "ss_bm25_upload_positions")
        self.indexed_doc_count = int(n_docs)
        self.lexical_field_count = 1
        self._df_cache.clear()

    def upload_lexical_fields(self, n_docs, doclen_bytes_fields, boost, term_offsets, doc_ids, field_ids, tfs, positions=None):
        """several indexed fields (BM25F): doclen [n_fields][n_docs], postings (doc, field, tf) sorted by (doc, field) per term
        positions: for every (term, doc, field) entry in that order its tf positions inside the field -- phrase queries only"""
        dl = np.ascontiguousarray(doclen_bytes_fields, np.uint8)
        b = None if boost is None else np.ascontiguousarray(boost, np.float32)
        off = np.ascontiguousarray(term_offsets, np.uint64)
        d = np.ascontiguousarray(doc_ids, np.uint32)
        f = np.ascontiguousarray(field_ids, np.uint8)
        t = np.ascontiguousarray(tfs, np.uint16)
        if positions is not None:
            ps = np.ascontiguousarray(positions, np.uint16)
            N.check(N.lib().ss_bm25_upload_fields_positions(self._h, int(n_docs), dl.shape[0], N.ptr(dl.reshape(-1), N.u8p),
                                                            N.ptr(b, N.f32p), len(off) - 1, N.ptr(off, N.u64p), N.ptr(d, N.u32p),
                                                            N.ptr(f, N.u8p), N.ptr(t, N.u16p), N.ptr(ps, N.u16p), len(ps)),
                    "ss_bm25_upload_fields_positions")
        else:
            N.check(N.lib().ss_bm25_upload_fields(self._h, int(n_docs), dl.shape[0], N.ptr(dl.reshape(-1), N.u8p), N.ptr(b, N.f32p),
                                                  len(off) - 1, N.ptr(off, N.u64p), N.ptr(d, N.u32p), N.ptr(f, N.u8p), N.ptr(t, N.u16p)),
                    "ss_bm25_upload_fields")
        self.indexed_doc_count = int(n_docs)
        self.lexical_field_count = int(dl.shape[0])
        self._df_cache.clear()

    def upload_ref_blocks(self, n_docs, doclen_bytes, term_blocks):
        """term_blocks: per term a list of (block_id, compression_type_pointer, posting_count, pointer_pivot_p_docid,
        key_body_bytes) in the reference's in-RAM format (ss_bm25_upload_ref_blocks)"""
        dl = np.ascontiguousarray(doclen_bytes, np.uint8)
        flat = [b for tb in term_blocks for b in tb]
        arr = (N.RefBlock * max(len(flat), 1))()
        keep = []
        for i, (bid, ctp, cnt, pivot, body) in enumerate(flat):
            buf = np.frombuffer(bytes(body), np.uint8).copy()
            keep.append(buf)
            arr[i] = N.RefBlock(bid, ctp, cnt - 1, pivot, buf.ctypes.data, len(buf))
        offs = np.zeros(len(term_blocks) + 1, np.uint64)
        offs[1:] = np.cumsum([len(tb) for tb in term_blocks])
        N.check(N.lib().ss_bm25_upload_ref_blocks(self._h, int(n_docs), N.ptr(dl, N.u8p), len(term_blocks), N.ptr(offs, N.u64p),
                                                  C.cast(arr, C.c_void_p)), "ss_bm25_upload_ref_blocks")
        self.indexed_doc_count = int(n_docs)
        self.lexical_field_count = 1
        self._df_cache.clear()

    def append_level(self, level, level_doclen, term_offsets, doc_ids, tfs, positions=None, npos=None):
        """one committed 65 536-doc level (commit.rs:142-148): its docs' length bytes and per term (shard-local doc id, tf); level =
        levels committed so far (append) or the last one (re-commit of a partial level).
        positions: every posting's positions in CSR order (phrase queries after commits); npos: their number per posting where not tf"""
        dl = np.ascontiguousarray(level_doclen, np.uint8)
        off = np.ascontiguousarray(term_offsets, np.uint64)
        d = np.ascontiguousarray(doc_ids, np.uint32)
        t = np.ascontiguousarray(tfs, np.uint16)
        if positions is not None:
            ps = np.ascontiguousarray(positions, np.uint16)
            npc = None if npos is None else np.ascontiguousarray(npos, np.uint16)
            N.check(N.lib().ss_bm25_append_level_positions(self._h, int(level), len(dl), N.ptr(dl, N.u8p), len(off) - 1, N.ptr(off, N.u64p), N.ptr(d, N.u32p),
                                                           N.ptr(t, N.u16p), N.ptr(npc, N.u16p), N.ptr(ps, N.u16p), len(ps)), "ss_bm25_append_level_positions")
        else:
            N.check(N.lib().ss_bm25_append_level(self._h, int(level), len(dl), N.ptr(dl, N.u8p), len(off) - 1, N.ptr(off, N.u64p), N.ptr(d, N.u32p),
                                                 N.ptr(t, N.u16p)), "ss_bm25_append_level")
        self.indexed_doc_count = int(level) * 65536 + len(dl)
        self.lexical_field_count = 1
        self._df_cache.clear()

    def append_level_fields(self, level, level_doclen, boost, term_offsets, doc_ids, fields, tfs):
        """one committed level of an image with SEVERAL indexed fields (ss_bm25_append_level_fields): level_doclen [n_fields][n_level_docs],
        entries (doc, field, tf) sorted by (doc, field) inside a term"""
        dl = np.ascontiguousarray(level_doclen, np.uint8)
        assert dl.ndim == 2
        off = np.ascontiguousarray(term_offsets, np.uint64)
        d = np.ascontiguousarray(doc_ids, np.uint32)
        f = np.ascontiguousarray(fields, np.uint8)
        t = np.ascontiguousarray(tfs, np.uint16)
        b = None if boost is None else np.ascontiguousarray(boost, np.float32)
        N.check(N.lib().ss_bm25_append_level_fields(self._h, int(level), dl.shape[1], dl.shape[0], N.ptr(dl.reshape(-1), N.u8p), N.ptr(b, N.f32p), len(off) - 1,
                                                    N.ptr(off, N.u64p), N.ptr(d, N.u32p), N.ptr(f, N.u8p), N.ptr(t, N.u16p)), "ss_bm25_append_level_fields")
        self.indexed_doc_count = int(level) * 65536 + dl.shape[1]
        self.lexical_field_count = dl.shape[0]
        self._df_cache.clear()

    def commit_level(self, level, level_doclen, term_offsets, doc_ids, tfs, n_dense_terms=None, positions=None, npos=None):
        """one commit as the seam sees it (commit.rs:142-148): the level's postings of ALL known terms in id order --
        ids below n_dense_terms belong to the dense image (ss_bm25_append_level), the others are the sparse tier's lists
        (ss_bm25_append_sparse_level; new rare terms simply extend the id range).  n_dense_terms None: no tier, every term is dense."""
        off = np.ascontiguousarray(term_offsets, np.uint64)
        nt = len(off) - 1
        nd = nt if n_dense_terms is None else int(n_dense_terms)
        if nd > nt:
            raise ValueError("fewer terms than the dense image holds")
        d = np.ascontiguousarray(doc_ids, np.uint32)
        t = np.ascontiguousarray(tfs, np.uint16)
        cut = int(off[nd])
        p_cut = None
        if positions is not None:
            cnt = t if npos is None else np.ascontiguousarray(npos, np.uint16)
            p_cut = int(cnt[int(off[0]):cut].astype(np.int64).sum())
        npd = None if npos is None else npos[:cut]
        self.append_level(level, level_doclen, off[:nd + 1], d[:cut], t[:cut], None if positions is None else positions[:p_cut], npd)
        if nd < nt:
            self.append_sparse_level(level, off[nd:] - off[nd], d[cut:], t[cut:], None if positions is None else positions[p_cut:],
                                     None if npos is None else npos[cut:])

    def incremental_info(self):
        """(levels, bytes of the raw postings kept for rebuilds, ms of the last append, of which the device rebuild)"""
        nl, rb, a, b = C.c_uint32(), C.c_uint64(), C.c_double(), C.c_double()
        N.check(N.lib().ss_bm25_incremental_info(self._h, C.byref(nl), C.byref(rb), C.byref(a), C.byref(b)), "ss_bm25_incremental_info")
        return int(nl.value), int(rb.value), float(a.value), float(b.value)

    def upload_index_bin(self, ix: "IndexBin", boost=None, positions=False):
        """boost: schema boost per indexed field (several fields only; schema.json)
        positions: also decode every posting's positions from the file (phrase queries; SingleTerm keys)"""
        b = None if boost is None else np.ascontiguousarray(boost, np.float32)
        if positions:
            N.check(N.lib().ss_bm25_upload_index_bin_fields_positions(self._h, ix._h, N.ptr(b, N.f32p)),
                    "ss_bm25_upload_index_bin_positions")
        else:
            N.check(N.lib().ss_bm25_upload_index_bin_fields(self._h, ix._h, N.ptr(b, N.f32p)), "ss_bm25_upload_index_bin")
        self.indexed_doc_count = int(ix.indexed_doc_count)
        self.lexical_field_count = int(ix.indexed_field_count)
        self._df_cache.clear()

    def upload_vector_bin(self, data, dim, i8=False, use_record_scale=False):
        """vector.bin as the reference writes it: f32 records, or Precision::I8 records (i8=True)"""
        buf = np.frombuffer(bytes(data), np.uint8)
        if i8:
            N.check(N.lib().ss_vec_upload_vector_bin_i8(self._h, buf.ctypes.data, len(buf), int(dim), 1 if use_record_scale else 0),
                    "ss_vec_upload_vector_bin_i8")
        else:
            N.check(N.lib().ss_vec_upload_vector_bin(self._h, buf.ctypes.data, len(buf), int(dim)), "ss_vec_upload_vector_bin")
        self.vector_precision = "i8" if i8 else "f32"
        n, d = C.c_uint64(), C.c_uint32()
        N.check(N.lib().ss_vec_info(self._h, C.byref(n), C.byref(d)), "ss_vec_info")
        self.vector_count, self.dim = n.value, d.value

    def set_clusters(self, level_clusters, child_count):
        """cluster structure of rows uploaded in the reference's order (level after level, cluster after cluster, the medoid
        first; vector.rs:1066-1094): clusters per level, records per cluster.  upload_vector_bin keeps the file's own."""
        lc = np.ascontiguousarray(level_clusters, np.uint32)
        cc = np.ascontiguousarray(child_count, np.uint32)
        if int(lc.sum()) != len(cc):
            raise ValueError("child_count must have one entry per cluster")
        N.check(N.lib().ss_vec_set_clusters(self._h, len(lc), N.ptr(lc, N.u32p), N.ptr(cc, N.u32p)), "ss_vec_set_clusters")

    def set_fields(self, row_field):
        """VectorHeader.field_id of every record (rows uploaded as arrays; upload_vector_bin reads the file's own)"""
        rf = np.ascontiguousarray(row_field, np.uint16)
        N.check(N.lib().ss_vec_set_fields(self._h, len(rf), rf.ctypes.data), "ss_vec_set_fields")

    def cluster_info(self):
        nl, nc = C.c_uint32(), C.c_uint32()
        N.check(N.lib().ss_vec_cluster_info(self._h, C.byref(nl), C.byref(nc)), "ss_vec_cluster_info")
        return nl.value, nc.value

    def set_deleted(self, doc_ids):
        """delete_hashset of the shard (index.rs:1594): replaces the tombstone set; also takes delete.bin's bytes"""
        if isinstance(doc_ids, (bytes, bytearray, memoryview)):
            doc_ids = np.frombuffer(bytes(doc_ids), "<u8")
        ids = np.ascontiguousarray(doc_ids, np.uint64)
        N.check(N.lib().ss_set_deleted(self._h, N.ptr(ids, N.u64p) if len(ids) else None, len(ids)), "ss_set_deleted")
        self._n_deleted = len(ids)

    def synth_partition(self, shard_id, n_shards):
        """the following synth_* calls build shard `shard_id` of `n_shards` of one generator stream (doc g -> shard g % S)"""
        N.check(N.lib().ss_synth_set_partition(self._h, int(shard_id), int(n_shards)), "ss_synth_set_partition")

    def synth_lexical(self, seed, n_docs, thresh32, len_table1024):
        th = np.ascontiguousarray(thresh32, np.uint32)
        tab = np.ascontiguousarray(len_table1024, np.uint8)
        assert tab.size == 1024
        N.check(N.lib().ss_bm25_synth(self._h, int(seed), int(n_docs), len(th), N.ptr(th, N.u32p), N.ptr(tab, N.u8p)),
                "ss_bm25_synth")
        self.indexed_doc_count = int(n_docs)
        self.lexical_field_count = 1
        self._df_cache.clear()

    def set_vector_similarity(self, similarity):
        """VectorSimilarity of the vector image: "dot" (Dot and Cosine: dot product) or "euclidean" (a record's similarity is
        MINUS its squared distance to the query, vector_similarity.rs:257-345); before the upload"""
        euclid = str(similarity).lower().startswith("euclid")
        N.check(N.lib().ss_vec_set_similarity(self._h, N.SIM_EUCLIDEAN if euclid else N.SIM_DOT), "ss_vec_set_similarity")
        self.vector_euclidean = euclid

    def set_row_norms(self, row_norm):
        """VectorHeader.norm of the i8 records (Euclidean + ScalarQuantizationI8: euclidean_i8_quantized)"""
        r = np.ascontiguousarray(row_norm, np.float32)
        N.check(N.lib().ss_vec_set_row_norms(self._h, len(r), N.ptr(r, N.f32p)), "ss_vec_set_row_norms")

    def upload_vectors(self, rows, row_doc_ids=None):
        r = np.ascontiguousarray(rows, np.float32)
        ids = None if row_doc_ids is None else np.ascontiguousarray(row_doc_ids, np.uint32)
        N.check(N.lib().ss_vec_upload(self._h, r.shape[0], r.shape[1], N.ptr(r, N.f32p), N.ptr(ids, N.u32p)), "ss_vec_upload")
        self.vector_count, self.dim = r.shape
        self.vector_precision = "f32"

    def upload_vectors_i8(self, rows_i8, row_scale=None, row_doc_ids=None):
        """Precision::I8 records: i8 components (+ VectorHeader.scale per record for ScalarQuantizationI8 with Dot)"""
        r = np.ascontiguousarray(rows_i8, np.int8)
        sc = None if row_scale is None else np.ascontiguousarray(row_scale, np.float32)
        ids = None if row_doc_ids is None else np.ascontiguousarray(row_doc_ids, np.uint32)
        N.check(N.lib().ss_vec_upload_i8(self._h, r.shape[0], r.shape[1], r.ctypes.data, N.ptr(sc, N.f32p), N.ptr(ids, N.u32p)),
                "ss_vec_upload_i8")
        self.vector_count, self.dim = r.shape
        self.vector_precision = "i8"

    def synth_vectors_i8(self, seed, n_rows, dim):
        N.check(N.lib().ss_vec_synth_i8(self._h, int(seed), int(n_rows), int(dim)), "ss_vec_synth_i8")
        self.vector_count, self.dim = int(n_rows), int(dim)
        self.vector_precision = "i8"

    def read_rows_i8(self, r0, n):
        out = np.empty((n, self.dim), np.int8)
        N.check(N.lib().ss_vec_read_rows_i8(self._h, int(r0), int(n), out.ctypes.data), "ss_vec_read_rows_i8")
        return out

    def search_vector_batch_i8(self, queries_i8, k, query_scale=None, similarity_threshold_raw=None, ann_mode=None,
                               with_clusters=False, field_filter=None, query_norm=None, with_observed=False):
        """scores = dot_i8 as f32 (* query_scale * embedding_scale with scales): vector_similarity.rs:1011-1016, 1754-1758;
        under Euclidean -euclidean_i8, or -euclidean_i8_quantized with the scales and norms (query_norm per query)"""
        qv = np.ascontiguousarray(queries_i8, np.int8)
        if qv.ndim == 1:
            qv = qv[None, :]
        if qv.shape[1] != self.dim:
            raise ValueError("query dimension mismatch")
        nq = qv.shape[0]
        qs = None if query_scale is None else np.ascontiguousarray(query_scale, np.float32)
        doc = np.empty((nq, k), np.uint32)
        score = np.empty((nq, k), np.float32)
        cnt = np.empty(nq, np.uint32)
        tot = np.empty(nq, np.uint64)
        thr = N.FLT_MIN_NEG if similarity_threshold_raw is None else float(similarity_threshold_raw)
        ncl = np.zeros(nq * (3 if with_observed else 1), np.uint32)
        mode = _vector_options(ann_mode, field_filter, self.vector_euclidean, with_observed)
        qn = None if query_norm is None else np.ascontiguousarray(query_norm, np.float32)
        N.check(N.lib().ss_vec_search_i8_euclid(self._h, nq, qv.ctypes.data, N.ptr(qs, N.f32p), N.ptr(qn, N.f32p), k, thr,
                                                None if mode is None else C.addressof(mode), N.ptr(doc, N.u32p),
                                                N.ptr(score, N.f32p), N.ptr(cnt, N.u32p), N.ptr(tot, N.u64p), N.ptr(ncl, N.u32p)),
                "ss_vec_search_i8_euclid")
        ncl, obs = _split_clusters(ncl, nq, with_observed)
        if with_observed:
            return doc, score, cnt, tot, ncl, obs
        return (doc, score, cnt, tot, ncl) if with_clusters else (doc, score, cnt, tot)

    def synth_vectors(self, seed, n_rows, dim):
        N.check(N.lib().ss_vec_synth(self._h, int(seed), int(n_rows), int(dim)), "ss_vec_synth")
        self.vector_count, self.dim = int(n_rows), int(dim)
        self.vector_precision = "f32"

    def read_rows(self, r0, n):
        out = np.empty((n, self.dim), np.float32)
        N.check(N.lib().ss_vec_read_rows(self._h, int(r0), int(n), N.ptr(out, N.f32p)), "ss_vec_read_rows")
        return out

    def set_probe_budget(self, max_bytes):
        """bytes of probe index the NEXT image build may spend (rows go to the longest posting lists first); 0 = default"""
        N.check(N.lib().ss_bm25_set_probe_budget(self._h, int(max_bytes)), "ss_bm25_set_probe_budget")

    def terms_probed(self, terms):
        t = np.ascontiguousarray(terms, np.uint32)
        out = np.zeros(len(t), np.uint8)
        N.check(N.lib().ss_bm25_term_probed(self._h, len(t), N.ptr(t, N.u32p), N.ptr(out, N.u8p)), "ss_bm25_term_probed")
        return out.astype(bool)

    def set_strategy(self, strategy):
        """N.BM25_AUTO / BM25_EXHAUSTIVE / BM25_PRUNED (ss_bm25_set_strategy); both strategies return identical results"""
        N.check(N.lib().ss_bm25_set_strategy(self._h, int(strategy)), "ss_bm25_set_strategy")

    def lexical_info(self):
        nd, av, nt, npost = C.c_uint64(), C.c_float(), C.c_uint32(), C.c_uint64()
        N.check(N.lib().ss_bm25_info(self._h, C.byref(nd), C.byref(av), C.byref(nt), C.byref(npost)), "ss_bm25_info")
        return dict(n_docs=nd.value, avgdl=av.value, n_terms=nt.value, n_postings=npost.value)

    def posting_count(self, terms):
        terms = np.ascontiguousarray(terms, np.uint32)
        out = np.empty(len(terms), np.uint64)
        N.check(N.lib().ss_bm25_term_df(self._h, len(terms), N.ptr(terms, N.u32p), N.ptr(out, N.u64p)), "ss_bm25_term_df")
        return out

    def algorithmic_bytes(self, queries, totals, k, n_docs=None):
        """SURVEY 8(d)'s algorithmic bytes of a batch (measurement only): per query sum_t df_t * (2 B doc id + 1 B tf) + 1 B per scored
        candidate (`totals`: the exact result_count_total of every query -- size of the union / intersection) + 4 B per (term, 65 536-doc
        block) + 8 B * k.  df = the docs holding the term in any indexed field (dense or sparse tier)."""
        n_docs = int(n_docs if n_docs is not None else self.lexical_info()["n_docs"])
        n_blocks = (n_docs + 65535) // 65536
        total = 0.0
        for q, cand in zip(queries, totals):
            nt = int(q["n_terms"])
            df = self.posting_count([int(t) for t in q["term"][:nt]])
            total += float(df.sum()) * 3.0 + float(cand) + 4.0 * n_blocks * nt + 8.0 * k
        return total

    # ---- query construction: term resolution + idf stay on the host (search.rs:3066-3358)
    def make_queries(self, term_lists: Sequence[Sequence[int]], query_types, not_lists=None, idf_of=None, field_filter=None):
        """query_list (+ not_query_list: the "-term" operands, add_result.rs:3440-3497) of each query -> ss_bm25_query.
        idf_of: {term id: idf} for the terms whose idf is not that of their own list -- the component terms of an n-gram
        key (IndexBin.terms_of_key: idf_ngram_i from the component term's posting count).
        field_filter: indexed field ids every term must occur in one of (several indexed fields; intersections and
        single-term queries, add_result.rs:3124-3136).
        An N-GRAM key among a query's terms is written as the tuple of its component term ids (IndexBin.terms_of_key, with their
        idf_ngram_i in idf_of): all components are scored; in a Phrase the key is ONE entry at its first place -- its first
        component, which carries the key's positions -- spanning len(tuple) places (search.rs:3305-3328)."""
        fmask = 0
        for f in field_filter or ():
            fmask |= 1 << int(f)
        if fmask >> 16:
            raise ValueError("field filter: indexed field ids 0..15")
        nq = len(term_lists)
        if not_lists is None:
            not_lists = [()] * nq
        if isinstance(query_types, (int, QueryType)):
            query_types = [query_types] * nq
        q = np.zeros(nq, N.BM25_QUERY_DTYPE)
        def _flat(tl):
            for t in tl:
                if isinstance(t, (tuple, list)):
                    yield from (int(x) for x in t)
                else:
                    yield int(t)
        flat = np.fromiter((t for tl in list(term_lists) + list(not_lists) for t in _flat(tl)), np.uint32)
        uniq = np.unique(flat)
        missing = [int(t) for t in uniq if int(t) not in self._df_cache]
        if missing:
            for t, df in zip(missing, self.posting_count(missing)):
                self._df_cache[t] = int(df)
        for i, (tl, qt, nl) in enumerate(zip(term_lists, query_types, not_lists)):
            entries = [tuple(int(x) for x in t) if isinstance(t, (tuple, list)) else (int(t),) for t in tl]
            tl = list(dict.fromkeys(_flat(tl)))  # unique_terms, search.rs:3023
            if int(qt) == int(QueryType.Phrase):  # non_unique_query_list: the entries in order, each naming its unique term
                places = sum(len(e) for e in entries)
                if len(entries) < 2:
                    qt = QueryType.Intersection  # a one-entry phrase is a term query (search.rs:3544)
                elif places > N.SS_MAX_PHRASE:
                    raise ValueError("a phrase of at most %d words" % N.SS_MAX_PHRASE)
                else:
                    q["phrase_len"][i] = places
                    at = 0
                    for e in entries:
                        q["phrase_seq"][i, at] = tl.index(e[0])
                        for x in range(1, len(e)):
                            q["phrase_seq"][i, at + x] = N.SS_PHRASE_SKIP  # a place inside the n-gram key
                        at += len(e)
            nl = [t for t in dict.fromkeys(_flat(nl)) if t not in tl]
            if not 1 <= len(tl):
                raise ValueError("a query needs at least one term")
            if len(tl) + len(nl) > N.SS_MAX_QUERY_TERMS:
                # (not an invalid query: the crate answers it -- union.rs:233-259 -- and this library's query record does not hold it)
                raise N.SeekStormHipError(N.SS_ENOTSUP, "more than %d unique terms (NOT terms included): the host's own dispatch" % N.SS_MAX_QUERY_TERMS)
            q["n_terms"][i] = len(tl)
            q["op"][i] = int(qt) | (len(nl) << 8) | (fmask << 16)
            for j, t in enumerate(tl):
                q["term"][i, j] = t
                q["idf"][i, j] = idf_of[t] if idf_of and t in idf_of else idf_f32(self.indexed_doc_count, self._df_cache[t])
            for j, t in enumerate(nl):
                q["term"][i, len(tl) + j] = t
        return q

    def upload_facets(self, records, record_size=None):
        """facet.bin: one record of facets_size_sum bytes per doc (bytes, or a [n_docs][record_size] uint8 array)"""
        if isinstance(records, (bytes, bytearray, memoryview)):
            r = np.frombuffer(bytes(records), np.uint8).reshape(-1, int(record_size))
        else:
            r = np.ascontiguousarray(records, np.uint8)
        N.check(N.lib().ss_facet_upload(self._h, r.shape[0], r.shape[1], r.ctypes.data), "ss_facet_upload")

    @staticmethod
    def facet_filters(filters):
        """[(offset, type, lo, hi)] for numeric facets -- passes iff lo <= value < hi, Rust's Range --,
        [(offset, "string16" | "string32", [ids])] or [(offset, "point", (lat, lon), lo, hi, "km" | "miles")] -- the distance to
        the base point inside [lo, hi) -> ss_facet_filter array (FacetFilter / FilterSparse, search.rs:735-)"""
        arr = (N.FacetFilterC * max(len(filters), 1))()
        keep = []
        arr._keep = keep  # the id arrays live as long as the filter array
        for i, f in enumerate(filters):
            off, ty = int(f[0]), f[1]
            arr[i].offset, arr[i].type = off, N.FACET_TYPES[ty]
            if ty == "point":  # (offset, "point", (lat, lon), lo, hi, unit[, flags]): distance to the base inside [lo, hi)
                base = np.array([f[2][0], f[2][1]], np.float64).view(np.uint32)
                for j in range(4):
                    arr[i].values[j] = int(base[j])
                rng_ = np.array([f[3], f[4]], np.float64).view(np.uint64)
                arr[i].lo, arr[i].hi, arr[i].n_values = int(rng_[0]), int(rng_[1]), N.POINT_UNITS[f[5]]
                arr[i].reserved = int(f[6]) if len(f) > 6 else 0
            elif ty.startswith("string"):
                ids = [int(x) for x in f[2]]
                if len(ids) > 8:  # any number of ids (what a StringSet filter resolves to): a host array behind lo / hi
                    ext = np.ascontiguousarray(ids, np.uint32)
                    keep.append(ext)
                    arr[i].n_values, arr[i].lo, arr[i].hi = N.FACET_IDS_EXTERN, ext.ctypes.data, len(ext)
                else:
                    arr[i].n_values = len(ids)
                    for j, v in enumerate(ids):
                        arr[i].values[j] = v
            elif len(f) > 4 and f[4] == "bits":  # (offset, type, lo bits, hi bits, "bits", flags): the pivots of a result sort
                arr[i].lo, arr[i].hi, arr[i].reserved = int(f[2]), int(f[3]), int(f[5])
            else:
                dt = {"f32": np.float32, "f64": np.float64}.get(ty)
                if dt is not None:
                    bits = np.array([f[2], f[3]], dt).view(np.uint32 if ty == "f32" else np.uint64)
                    arr[i].lo, arr[i].hi = int(bits[0]), int(bits[1])
                else:
                    arr[i].lo, arr[i].hi = int(f[2]) & 0xFFFFFFFFFFFFFFFF, int(f[3]) & 0xFFFFFFFFFFFFFFFF
        return arr, len(filters)

    # ---- result sort (search.rs ResultSort; ordering min_heap.rs:574-1050)
    _FACET_BITS = {"u8": 8, "u16": 16, "u32": 32, "u64": 64, "i8": 8, "i16": 16, "i32": 32, "i64": 64, "f32": 32, "f64": 64}

    @staticmethod
    def _facet_order_key(bits, ty, descending):
        """stored bits -> an integer that orders like the value (larger = better under the sort)"""
        nb = Shard._FACET_BITS[ty]
        mask, top = (1 << nb) - 1, 1 << (nb - 1)
        k = int(bits) & mask
        if ty[0] == "i":
            k ^= top
        elif ty[0] == "f":
            k = (~k & mask) if (k & top) else (k | top)
        return k if descending else (~k & mask)

    @staticmethod
    def _facet_filter_bits(v, ty):
        """stored bits (zero-extended) -> the form ss_facet_filter compares: signed integers sign-extended to 64 bits"""
        nb = Shard._FACET_BITS[ty]
        if ty[0] == "i" and nb < 64 and (int(v) >> (nb - 1)) & 1:
            return int(v) | (0xFFFFFFFFFFFFFFFF & ~((1 << nb) - 1))
        return int(v)

    @staticmethod
    def _facet_type_range(ty):
        """(smallest, largest) value of the type in ss_facet_filter's form"""
        nb = Shard._FACET_BITS[ty]
        if ty[0] == "u":
            return 0, (1 << nb) - 1
        if ty[0] == "i":
            return Shard._facet_filter_bits(1 << (nb - 1), ty), (1 << (nb - 1)) - 1
        return (0xFF800000, 0x7F800000) if nb == 32 else (0xFFF0000000000000, 0x7FF0000000000000)  # -inf, +inf

    @staticmethod
    def _point(base, unit):
        return N.FacetPointC(float(base[0]), float(base[1]), N.POINT_UNITS[unit], 0)

    @staticmethod
    def string_facet_rank_column(records, facet_offset, facet_type, strings):
        """Result sort by a String16 / String32 facet (min_heap.rs:860-897, 939-976): the reference compares the STRINGS of the
        two docs' value ids (Rust String order = byte-wise UTF-8).  The device sorts numbers, so the host appends one derived
        u32 column to the facet.bin records before ss_facet_upload: rank[id] of the id's string in that order (equal strings
        share a rank) -- a sort by the string facet is then a sort by this column.  The ranks are as old as the image: a commit
        that adds strings rebuilds image and column together.  strings[id] = the facet value of id (facet.json).
        -> (records with the column appended [n_docs][record_size + 4], the column's offset)"""
        r = np.ascontiguousarray(records, np.uint8)
        width = {"string16": 2, "string32": 4}[facet_type]
        ids = np.zeros(r.shape[0], np.uint32)
        for b in range(width):
            ids |= r[:, facet_offset + b].astype(np.uint32) << np.uint32(8 * b)
        keys = [str(x).encode("utf-8") for x in strings]
        order = sorted(set(keys))
        rank_of = {kb: i for i, kb in enumerate(order)}
        rank = np.array([rank_of[kb] for kb in keys], np.uint32)
        col = rank[np.minimum(ids, len(rank) - 1)]
        out = np.concatenate([r, col.astype("<u4").view(np.uint8).reshape(-1, 4)], axis=1)
        return np.ascontiguousarray(out), r.shape[1]

    def facet_kth(self, query, facet_offset, facet_type, descending, k, facet_filter=None, base=None):
        """the pivot of a result sort: (stored bits of the k-th best value among the query's matches, matches strictly better,
        matches equal, all matches) -- ss_bm25_facet_kth; a Point facet (base = (lat, lon)): the f64 bits of the k-th best
        simplified_distance to the base -- ss_bm25_facet_kth_point"""
        farr, nf = self.facet_filters(facet_filter) if facet_filter else (None, 0)
        q = np.ascontiguousarray(query[:1])
        v, nb, ne, tot = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        fp = None if farr is None else C.cast(farr, C.c_void_p)
        if facet_type == "point":
            pt = self._point(base, "sortkey")
            N.check(N.lib().ss_bm25_facet_kth_point(self._h, q.ctypes.data, nf, fp, int(facet_offset), C.byref(pt), 1 if descending else 0,
                                                    int(k), C.byref(v), C.byref(nb), C.byref(ne), C.byref(tot)), "ss_bm25_facet_kth_point")
        else:
            N.check(N.lib().ss_bm25_facet_kth(self._h, q.ctypes.data, nf, fp, int(facet_offset), N.FACET_TYPES[facet_type],
                                              1 if descending else 0, int(k), C.byref(v), C.byref(nb), C.byref(ne), C.byref(tot)),
                    "ss_bm25_facet_kth")
        return v.value, nb.value, ne.value, tot.value

    def facet_values(self, doc_ids, facet_offset, facet_type):
        d = np.ascontiguousarray(doc_ids, np.uint32)
        out = np.zeros(len(d), np.uint64)
        N.check(N.lib().ss_facet_values(self._h, len(d), N.ptr(d, N.u32p), int(facet_offset), N.FACET_TYPES[facet_type],
                                        N.ptr(out, N.u64p)), "ss_facet_values")
        return out

    def facet_point_distances(self, doc_ids, facet_offset, base, unit="km"):
        """f64 distances of the docs' Point facet to base = (lat, lon): euclidian_distance in km / miles (geo_search.rs:115-124),
        or unit "sortkey": simplified_distance, what a sort by distance compares (geo_search.rs:82-87)"""
        d = np.ascontiguousarray(doc_ids, np.uint32)
        out = np.zeros(len(d), np.uint64)
        pt = self._point(base, unit)
        N.check(N.lib().ss_facet_point_distances(self._h, len(d), N.ptr(d, N.u32p), int(facet_offset), C.byref(pt), N.ptr(out, N.u64p)),
                "ss_facet_point_distances")
        return out.view(np.float64)

    def search_lexical_sorted_batch(self, queries, result_sort, k, facet_filter=None):
        """a BATCH of queries under result_sort = [(facet offset, type, descending[, base])] (numeric facets, Point facets by
        simplified_distance to base = (lat, lon)): ss_bm25_search_sorted -- the pivots of every sort field are found on the device, two
        searches under exclusion bitmaps and a compose kernel per query, one synchronisation per call.
        -> (doc [nq][k], score [nq][k], count [nq], total [nq])"""
        n = len(result_sort)
        arr = (N.ResultSortC * max(n, 1))()
        for i, sf in enumerate(result_sort):
            off, ty, desc = sf[:3]
            arr[i].facet_offset, arr[i].descending = int(off), 1 if desc else 0
            if ty == "point":
                arr[i].facet_type, arr[i].base_lat, arr[i].base_lon = N.FACET_TYPES["point"], float(sf[3][0]), float(sf[3][1])
            else:
                arr[i].facet_type = N.FACET_TYPES[ty]
        q = np.ascontiguousarray(queries)
        nq, kk = len(q), int(k)
        doc = np.full((nq, kk), N.SS_NO_DOC, np.uint32)
        score = np.zeros((nq, kk), np.float32)
        cnt = np.zeros(nq, np.uint32)
        tot = np.zeros(nq, np.uint64)
        farr, nf = self.facet_filters(facet_filter) if facet_filter else (None, 0)
        N.check(N.lib().ss_bm25_search_sorted(self._h, nq, q.ctypes.data_as(C.c_void_p), n, C.cast(arr, C.c_void_p), kk, nf,
                                              None if farr is None else C.cast(farr, C.c_void_p), N.ptr(doc, N.u32p), N.ptr(score, N.f32p),
                                              N.ptr(cnt, N.u32p), N.ptr(tot, N.u64p)), "ss_bm25_search_sorted")
        return doc, score, cnt, tot

    def search_lexical_sorted(self, query, result_sort, k, facet_filter=None):
        """ONE query (a 1-element make_queries array) under result_sort -> (doc ids, scores, total): ss_bm25_search_sorted"""
        if len(result_sort) > N.SS_MAX_SORT_FIELDS:
            return self.search_lexical_sorted_composed(query, result_sort, k, facet_filter)
        doc, score, cnt, tot = self.search_lexical_sorted_batch(query[:1], result_sort, k, facet_filter)
        return doc[0][:cnt[0]].copy(), score[0][:cnt[0]].copy(), int(tot[0])

    def search_lexical_sorted_composed(self, query, result_sort, k, facet_filter=None):
        """the same answer composed on the HOST from the older entry points (kept as the second route and as a cross-check):
        ONE query (a 1-element make_queries array) with result_sort = [(facet offset, type, descending[, base])], the reference's
        Vec<ResultSort> over numeric facets and Point facets (type "point", base = (lat, lon): by simplified_distance to the base,
        morton_ordering, geo_search.rs:90-108): the k best matches under (field 1, field 2, ..., score), each field ascending or
        descending, the score descending last (result_ordering_shard, min_heap.rs:574-1050) -> (doc ids, scores, total).
        Composed from ordinary searches around the sort's pivot: the k-th best value of the first field among the matches
        (ss_bm25_facet_kth) splits them into "strictly better" -- fewer than k docs, fetched by a search filtered to that
        range and ordered here by their values -- and "equal", of which the best by the REMAINING sort fields fill the rest."""
        flt = list(facet_filter or [])
        if len(flt) + len(result_sort) > N.SS_MAX_FACET_FILTERS:
            raise ValueError("at most %d facet filters + sort fields" % N.SS_MAX_FACET_FILTERS)
        q = np.ascontiguousarray(query[:1])
        total = [None]

        def topk(sorts, kk, filters):
            if kk == 0:
                return []
            if not sorts:
                doc, score, cnt, tot = self.search_lexical_batch(q, kk, ResultType.TopkCount, facet_filter=filters or None)
                if total[0] is None:
                    total[0] = int(tot[0])
                return [(int(doc[0][i]), float(score[0][i])) for i in range(int(cnt[0]))]
            off, ty, desc = sorts[0][:3]
            base = sorts[0][3] if ty == "point" else None
            v, n_better, n_equal, tot = self.facet_kth(q, off, ty, desc, kk, filters or None, base=base)
            if total[0] is None:
                total[0] = tot
            if tot == 0:
                return []
            out = []
            if ty == "point":  # the pivot is a distance: "better" / "equal" are distance filters on the sort key
                pv = float(np.array([v], np.uint64).view(np.float64)[0])
                better = ((off, ty, base, pv, np.inf, "sortkey", N.FACET_LO_EXCLUSIVE | N.FACET_HI_INCLUSIVE) if desc
                          else (off, ty, base, -np.inf, pv, "sortkey", 0))
                equal = (off, ty, base, pv, pv, "sortkey", N.FACET_HI_INCLUSIVE)
            else:
                vb = self._facet_filter_bits(v, ty)
                lo_all, hi_all = self._facet_type_range(ty)
                # strictly better than the pivot: above it for a descending sort, below it for an ascending one
                better = ((off, ty, vb, hi_all, "bits", N.FACET_LO_EXCLUSIVE | N.FACET_HI_INCLUSIVE) if desc
                          else (off, ty, lo_all, vb, "bits", 0))
                equal = (off, ty, vb, vb, "bits", N.FACET_HI_INCLUSIVE)
            if n_better:
                got = topk([], n_better, filters + [better])
                docs = [d for d, _ in got]
                keys = [[self._facet_order_key(x, "f64" if sf[1] == "point" else sf[1], sf[2]) for x in
                         (self.facet_point_distances(docs, sf[0], sf[3], "sortkey").view(np.uint64) if sf[1] == "point"
                          else self.facet_values(docs, sf[0], sf[1]))] for sf in sorts]
                order = sorted(range(len(got)), key=lambda i: tuple(-kk_[i] for kk_ in keys) + (-got[i][1], got[i][0]))
                out += [got[i] for i in order]
            if n_better < kk and n_equal:
                out += topk(sorts[1:], kk - n_better, filters + [equal])
            return out

        res = topk(list(result_sort), int(k), flt)
        return (np.array([d for d, _ in res], np.uint32), np.array([s_ for _, s_ in res], np.float32), int(total[0] or 0))

    def facet_count(self, query, facet_offset, facet_type, n_buckets=None, range_lower_bounds=None, facet_filter=None, base=None,
                    unit="km"):
        """query_facets of one query (facet_count, add_result.rs:484-640): histogram of the facet over the match set.
        String facets: n_buckets ids; numeric facets: ascending lower bounds of the ranges; Point facets (QueryFacet::Point):
        base = (lat, lon), unit, and the lower bounds of the DISTANCE ranges.  -> (counts [n_buckets], docs outside the
        buckets, match count)"""
        if facet_type == "point":
            bounds = np.asarray(range_lower_bounds, np.float64).view(np.uint64).copy()
            nb = len(bounds)
            out = np.zeros(nb + 1, np.uint64)
            tot = C.c_uint64()
            farr, nf = self.facet_filters(facet_filter) if facet_filter else (None, 0)
            q = np.ascontiguousarray(query[:1])
            pt = self._point(base, unit)
            N.check(N.lib().ss_bm25_facet_count_point(self._h, q.ctypes.data, nf, None if farr is None else C.cast(farr, C.c_void_p),
                                                      int(facet_offset), C.byref(pt), nb, N.ptr(bounds, N.u64p), N.ptr(out, N.u64p),
                                                      C.byref(tot)), "ss_bm25_facet_count_point")
            return out[:nb].copy(), int(out[nb]), tot.value
        if facet_type.startswith("string"):
            nb, bounds = int(n_buckets), None
        else:
            dt = {"f32": np.float32, "f64": np.float64}.get(facet_type)
            if dt is None:
                bounds = np.array([int(x) & 0xFFFFFFFFFFFFFFFF for x in range_lower_bounds], np.uint64)
            elif facet_type == "f32":
                bounds = np.asarray(range_lower_bounds, np.float32).view(np.uint32).astype(np.uint64)
            else:
                bounds = np.asarray(range_lower_bounds, np.float64).view(np.uint64).copy()
            nb = len(bounds)
        out = np.zeros(nb + 1, np.uint64)
        tot = C.c_uint64()
        farr, nf = self.facet_filters(facet_filter) if facet_filter else (None, 0)
        q = np.ascontiguousarray(query[:1])
        N.check(N.lib().ss_bm25_facet_count(self._h, q.ctypes.data, nf, None if farr is None else C.cast(farr, C.c_void_p),
                                            int(facet_offset), N.FACET_TYPES[facet_type], nb, N.ptr(bounds, N.u64p),
                                            N.ptr(out, N.u64p), C.byref(tot)), "ss_bm25_facet_count")
        return out[:nb].copy(), int(out[nb]), tot.value

    def fields_info(self):
        """(indexed fields, merged lists present, positions present) of the lexical image"""
        nf, mg, ps = (np.zeros(1, np.uint32) for _ in range(3))
        N.check(N.lib().ss_bm25_fields_info(self._h, N.ptr(nf, N.u32p), N.ptr(mg, N.u32p), N.ptr(ps, N.u32p)), "ss_bm25_fields_info")
        return int(nf[0]), bool(mg[0]), bool(ps[0])

    def mark_all_terms_frequent(self, queries, k):
        """The reference's all_terms_frequent condition (intersection.rs:198-209), evaluated where the reference evaluates
        it -- on the host, per query: indexed_doc_count > top_k << 8 and posting_count / indexed_doc_count >= 0.5 (f32) for
        every term of an intersection of several terms.  Returns the queries with SS_OP_ALL_TERMS_FREQUENT set where it
        holds (a copy if anything changed): such a query counts every match but ranks only docs whose every tf >= 10 (several
        indexed fields: the tf in the lowest field that holds the doc, add_result.rs:1595-1607)."""
        if self.indexed_doc_count <= (int(k) << 8):
            return queries
        cand = np.nonzero(((queries["op"] & 0xFF) == int(QueryType.Intersection)) & (queries["n_terms"] > 1) &
                          (((queries["op"] >> 16) & 0x7FFF) == 0))[0]  # not under a field filter, add_result.rs:3545
        if len(cand) == 0:
            return queries
        # vectorised: a 1000-query batch costs one df lookup for its unseen terms and a few array operations (the Python loop
        # it replaces cost more than the device call it precedes)
        nt = queries["n_terms"][cand].astype(np.int64)
        terms = queries["term"][cand].astype(np.int64)
        valid = np.arange(terms.shape[1])[None, :] < nt[:, None]
        uniq = np.unique(terms[valid])
        dfu = self.posting_count(uniq).astype(np.float32)  # one call: a host-side table lookup in the library
        freq_u = dfu / np.float32(self.indexed_doc_count) >= np.float32(0.5)  # f32 division, as the reference's
        if not freq_u.any():
            return queries
        freq = freq_u[np.minimum(np.searchsorted(uniq, terms), len(uniq) - 1)] | ~valid
        hit = cand[freq.all(axis=1)]
        if len(hit) == 0:
            return queries
        out = queries.copy()
        out["op"][hit] |= np.uint32(0x80000000)
        return out

    # ---- batched executors (one C-ABI call per batch)
    def search_lexical_batch(self, queries, k, result_type=ResultType.TopkCount, reference_shortcuts=True, facet_filter=None):
        """reference_shortcuts: apply all_terms_frequent where its condition holds, as the reference does.
        facet_filter: see facet_filters(); shared by the queries of the call (a filtered doc neither counts nor ranks)"""
        # a facet filter disables the shortcut (add_result.rs:2096-2100: all_terms_frequent && !phrase_query && !facet_filtered)
        if reference_shortcuts and result_type != ResultType.Count and not facet_filter and \
                (self.lexical_field_count == 1 or self.fields_info()[1]):  # several fields: over the merged lists
            queries = self.mark_all_terms_frequent(queries, k)
        nq = len(queries)
        kk = max(int(k), 1)
        doc = np.full((nq, kk), N.SS_NO_DOC, np.uint32)
        score = np.zeros((nq, kk), np.float32)
        cnt = np.zeros(nq, np.uint32)
        tot = np.zeros(nq, np.uint64)
        # (a batch may mix phrase queries with others: the library runs it as two and puts the answers back in place)
        farr, nf = self.facet_filters(facet_filter) if facet_filter else (None, 0)
        N.check(N.lib().ss_bm25_search_filtered(self._h, nq, queries.ctypes.data_as(C.c_void_p), int(k), int(result_type), nf,
                                                None if farr is None else C.cast(farr, C.c_void_p), N.ptr(doc, N.u32p),
                                                N.ptr(score, N.f32p), N.ptr(cnt, N.u32p), N.ptr(tot, N.u64p)),
                "ss_bm25_search_filtered")
        return doc, score, cnt, tot

    def search_vector_batch(self, query_vectors, k, similarity_threshold=None, ann_mode=None, with_clusters=False,
                            field_filter=None, with_observed=False):
        """with_observed: -> (..., observed_cluster_count, observed_vector_count) per query (vector.rs:421, 1394, 1510)"""
        qv = np.ascontiguousarray(query_vectors, np.float32)
        if qv.ndim == 1:
            qv = qv[None, :]
        if qv.shape[1] != self.dim:
            raise ValueError("query dimension mismatch")
        nq = qv.shape[0]
        doc = np.full((nq, k), N.SS_NO_DOC, np.uint32)
        score = np.zeros((nq, k), np.float32)
        cnt = np.zeros(nq, np.uint32)
        tot = np.zeros(nq, np.uint64)
        ncl = np.zeros(nq * (3 if with_observed else 1), np.uint32)
        mode = _vector_options(ann_mode, field_filter, self.vector_euclidean, with_observed)
        N.check(N.lib().ss_vec_search_ann(self._h, nq, N.ptr(qv, N.f32p), int(k), threshold_raw(similarity_threshold, self.vector_euclidean),
                                          None if mode is None else C.addressof(mode), N.ptr(doc, N.u32p), N.ptr(score, N.f32p),
                                          N.ptr(cnt, N.u32p), N.ptr(tot, N.u64p), N.ptr(ncl, N.u32p)), "ss_vec_search_ann")
        ncl, obs = _split_clusters(ncl, nq, with_observed)
        if with_observed:
            return doc, score, cnt, tot, ncl, obs
        return (doc, score, cnt, tot, ncl) if with_clusters else (doc, score, cnt, tot)

    # ---- the reference's per-shard seams (one query)
    def search_lexical_shard(self, query_terms, query_type_default=QueryType.Union, offset=0, length=10,
                             result_type=ResultType.TopkCount, strict=False, not_terms=(), field_filter=None,
                             facet_filter=None) -> ResultObject:
        """search.rs:2427-2442: field_filter = indexed field ids, facet_filter = see facet_filters()"""
        ro = ResultObject()
        try:
            uniq = list(dict.fromkeys(int(t) for t in query_terms))
            # a union of several terms under a field filter: per-term gating inside the scan kernels (<= 7 terms, round 3) or the
            # composition from the reference's own sub-queries BEHIND the ABI (8 .. 10 terms, a sparse-tier term; round 6); the
            # host-side composition below stays as a second route for the tests (compose_filtered_unions = True, <= 5 terms)
            if (field_filter and self.lexical_field_count > 1 and int(query_type_default) == int(QueryType.Union) and len(uniq) > 1
                    and self.compose_filtered_unions):
                return self._union_with_field_filter(uniq, offset, length, result_type, not_terms, field_filter, facet_filter)
            q = self.make_queries([query_terms], query_type_default, [not_terms], field_filter=field_filter)
            doc, score, cnt, tot = self.search_lexical_batch(q, offset + length, result_type, facet_filter=facet_filter)
        except Exception as e:
            if strict:
                raise
            ro.cpu_dispatch = isinstance(e, N.SeekStormHipError) and e.code == N.SS_ENOTSUP
            return ro
        n = int(cnt[0])
        ro.results = [Result(int(d), float(s), ResultSource.Lexical) for d, s in zip(doc[0, :n], score[0, :n])][offset:]
        ro.result_count = len(ro.results)
        ro.result_count_total = int(tot[0])
        return ro

    def _union_with_field_filter(self, terms, offset, length, result_type, not_terms, field_filter, facet_filter):
        """A union of several terms under a field filter.  The reference answers it through sub-queries: union_docid_3 queues
        the intersection of all terms and every subset one term shorter, down to pairs (union.rs:1330-1425), union_docid_2
        runs a pair as its intersection plus the two single terms (union.rs:1168-1305), the filter (add_result.rs:3124-3136)
        applies to the terms of the sub-query that finds the doc, and a doc found again keeps its better score
        (docid_hashset, min_heap.rs:1193-1260).  Every subset of the query's terms is thus tried as a filtered intersection
        and a doc ends with the best of them: the sum over its terms that occur in a listed field (all fields of those terms
        counted), a doc none of whose terms passes is no result.  The same here: the 2^n - 1 filtered intersections as ONE
        device batch, merged per doc by the maximum.  Totals as the reference reports them: two terms -> |pass(X) u pass(Y)|
        (union_docid_2's count), more -> the unfiltered union (union_scan counts a doc before the filter sees it,
        union.rs:552-553)."""
        n = len(terms)
        if n > 5:
            raise ValueError("a union of more than 5 terms under a field filter (2^n - 1 sub-queries)")
        k = offset + length
        subsets = [[terms[i] for i in range(n) if (m >> i) & 1] for m in range(1, 1 << n)]
        q = self.make_queries(subsets, QueryType.Intersection, [list(not_terms)] * len(subsets), field_filter=field_filter)
        rt = ResultType.TopkCount if result_type != ResultType.Topk else ResultType.Topk
        doc, score, cnt, tot = self.search_lexical_batch(q, max(k, 1), rt, facet_filter=facet_filter)
        best = {}
        for i in range(len(subsets)):
            for d, sc in zip(doc[i, :int(cnt[i])], score[i, :int(cnt[i])]):
                if float(sc) > best.get(int(d), -1.0):
                    best[int(d)] = float(sc)
        ranked = sorted(best.items(), key=lambda e: (-e[1], e[0]))[:k]
        ro = ResultObject()
        if result_type != ResultType.Count:
            ro.results = [Result(d, sc, ResultSource.Lexical) for d, sc in ranked][offset:]
        ro.result_count = len(ro.results)
        if result_type != ResultType.Topk:
            if n == 2:  # |A| + |B| - |A n B| over the filtered lists
                ro.result_count_total = int(tot[0]) + int(tot[1]) - int(tot[2])
            else:
                qu = self.make_queries([terms], QueryType.Union, [list(not_terms)])
                ro.result_count_total = int(self.search_lexical_batch(qu, 1, ResultType.Count, facet_filter=facet_filter)[3][0])
        else:
            ro.result_count_total = len(ranked)
        return ro

    def search_vector_shard(self, query_vector, length=10, similarity_threshold=None, strict=False,
                            ann_mode=None, field_filter=None) -> ResultObject:
        ro = ResultObject()
        try:
            if self.vector_precision == "i8":  # the query is quantised like the records (search.rs:1476-1490), threshold on the raw dot
                q8 = quantize_f32_to_i8(np.ascontiguousarray(query_vector, np.float32).reshape(1, -1))
                thr = None if similarity_threshold is None else threshold_raw(similarity_threshold, self.vector_euclidean)
                doc, score, cnt, tot, ncl, obs = self.search_vector_batch_i8(q8, length, similarity_threshold_raw=thr,
                                                                             ann_mode=ann_mode, field_filter=field_filter,
                                                                             with_observed=True)
            else:
                # (AnnMode::All on a shard without tombstones and without a field filter observes every record: no mode is
                # passed, and such calls coalesce behind the ABI)
                plain = ann_mode is None and not field_filter and not getattr(self, "_n_deleted", 0)
                r = self.search_vector_batch(query_vector, length, similarity_threshold, ann_mode=ann_mode, field_filter=field_filter,
                                             with_observed=not plain, with_clusters=True)
                doc, score, cnt, tot, ncl = r[:5]
                obs = r[5] if not plain else np.array([self.vector_count], np.uint64)
        except Exception:
            if strict:
                raise
            return ro
        n = int(cnt[0])
        ro.results = [Result(int(d), float(s), ResultSource.Vector) for d, s in zip(doc[0, :n], score[0, :n])]
        ro.result_count = n
        ro.result_count_total = int(tot[0])
        # TopK::push counts every record it is handed (vector.rs:421, 1510): those of the visited clusters that pass the field
        # filter and are not tombstoned -- counted on the device (SS_ANN_REPORT_OBSERVED)
        ro.observed_vector_count = int(obs[0])
        if ann_mode is None:
            ro.observed_cluster_count = max(self.cluster_info()[1], 1)  # AnnMode::All: every cluster (one when none is declared), as the C++ mirror
        else:
            ro.observed_cluster_count = int(ncl[0])  # vector.rs:1394
        return ro

    # ---- measurement hooks
    def reserve_vector_rows(self, n_rows_cap):
        """room for n_rows_cap rows, once, so that the appends that follow write in place (ss_vec_reserve_rows)"""
        N.check(N.lib().ss_vec_reserve_rows(self._h, int(n_rows_cap)), "ss_vec_reserve_rows")

    def append_vector_rows(self, rows, row_doc_ids=None, row_scale=None, row_norm=None, row_field=None, child_count=None):
        """one committed level of vector records behind the image's rows (ss_vec_append_rows): rows f32 [n][dim], or int8 for an i8
        image; the per-row arrays and the level's cluster child counts exactly as the image carries them"""
        r = np.ascontiguousarray(rows)
        i8 = r.dtype == np.int8
        if not i8:
            r = np.ascontiguousarray(r, np.float32)
        if r.ndim != 2 or r.shape[1] != self.dim:
            raise ValueError("rows must be [n][dim]")
        keep = [r]

        def p(a, dt):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dt)
            keep.append(a)
            return a.ctypes.data
        cc = None if child_count is None else np.ascontiguousarray(child_count, np.uint32)
        lv = N.VecLevelC(r.shape[0], r.ctypes.data, 1 if i8 else 0, 0 if cc is None else len(cc), p(row_doc_ids, np.uint32), p(row_scale, np.float32),
                         p(row_norm, np.float32), p(row_field, np.uint16), None if cc is None else cc.ctypes.data)
        N.check(N.lib().ss_vec_append_rows(self._h, C.byref(lv)), "ss_vec_append_rows")
        n, d = C.c_uint64(), C.c_uint32()
        N.check(N.lib().ss_vec_info(self._h, C.byref(n), C.byref(d)), "ss_vec_info")
        self.vector_count = n.value

    def append_sparse(self, term_offsets, doc_ids, tfs, positions=None, npos=None):
        """lists of RARE terms into the image's sparse tier (ss_bm25_append_sparse: plain sorted lists, no directory / probe rows);
        returns the term id of the first appended list -- the ids continue behind the dense terms.
        positions: every posting's positions in order (phrase queries naming a sparse term); npos: their number per posting where
        that is not the tf"""
        offs = np.ascontiguousarray(term_offsets, np.uint64)
        d = np.ascontiguousarray(doc_ids, np.uint32)
        t = np.ascontiguousarray(tfs, np.uint16)
        first = C.c_uint32()
        if positions is not None:
            ps = np.ascontiguousarray(positions, np.uint16)
            npc = None if npos is None else np.ascontiguousarray(npos, np.uint16)
            N.check(N.lib().ss_bm25_append_sparse_positions(self._h, len(offs) - 1, N.ptr(offs, N.u64p), N.ptr(d, N.u32p), N.ptr(t, N.u16p),
                                                            N.ptr(ps, N.u16p), len(ps), N.ptr(npc, N.u16p), C.byref(first)),
                    "ss_bm25_append_sparse_positions")
            self._df_cache.clear()
            return int(first.value)
        N.check(N.lib().ss_bm25_append_sparse(self._h, len(offs) - 1, N.ptr(offs, N.u64p), N.ptr(d, N.u32p), N.ptr(t, N.u16p), C.byref(first)),
                "ss_bm25_append_sparse")
        self._df_cache.clear()
        return int(first.value)

    def append_sparse_level(self, level, term_offsets, doc_ids, tfs, positions=None, npos=None):
        """the committed level's postings of the RARE terms of an image that grows by levels (ss_bm25_append_sparse_level, after
        append_level of its dense terms): list i continues sparse list i, further lists are new terms; a level brought before is replaced"""
        offs = np.ascontiguousarray(term_offsets, np.uint64)
        d = np.ascontiguousarray(doc_ids, np.uint32)
        t = np.ascontiguousarray(tfs, np.uint16)
        ps = None if positions is None else np.ascontiguousarray(positions, np.uint16)
        if ps is not None and len(ps) == 0:
            ps = np.zeros(1, np.uint16)  # (a non-null pointer says "this tier carries positions")
            n_ps = 0
        else:
            n_ps = 0 if ps is None else len(ps)
        npc = None if npos is None else np.ascontiguousarray(npos, np.uint16)
        N.check(N.lib().ss_bm25_append_sparse_level(self._h, int(level), len(offs) - 1, N.ptr(offs, N.u64p), N.ptr(d, N.u32p), N.ptr(t, N.u16p),
                                                    N.ptr(npc, N.u16p), N.ptr(ps, N.u16p), n_ps), "ss_bm25_append_sparse_level")
        self._df_cache.clear()

    def append_sparse_fields(self, term_offsets, doc_ids, field_ids, tfs, positions=None, npos=None):
        """... on an image with several indexed fields: entries (doc, field, tf) sorted by (doc, field) per term; the tier keeps the
        terms' merged lists (ss_bm25_append_sparse_fields); positions: every entry's positions inside its field"""
        offs = np.ascontiguousarray(term_offsets, np.uint64)
        d = np.ascontiguousarray(doc_ids, np.uint32)
        f = np.ascontiguousarray(field_ids, np.uint8)
        t = np.ascontiguousarray(tfs, np.uint16)
        first = C.c_uint32()
        if positions is not None:
            ps = np.ascontiguousarray(positions, np.uint16)
            npc = None if npos is None else np.ascontiguousarray(npos, np.uint16)
            N.check(N.lib().ss_bm25_append_sparse_fields_positions(self._h, len(offs) - 1, N.ptr(offs, N.u64p), N.ptr(d, N.u32p), N.ptr(f, N.u8p),
                                                                   N.ptr(t, N.u16p), N.ptr(ps, N.u16p), len(ps), N.ptr(npc, N.u16p),
                                                                   C.byref(first)), "ss_bm25_append_sparse_fields_positions")
            self._df_cache.clear()
            return int(first.value)
        N.check(N.lib().ss_bm25_append_sparse_fields(self._h, len(offs) - 1, N.ptr(offs, N.u64p), N.ptr(d, N.u32p), N.ptr(f, N.u8p),
                                                     N.ptr(t, N.u16p), C.byref(first)), "ss_bm25_append_sparse_fields")
        self._df_cache.clear()
        return int(first.value)

    def sparse_info(self):
        """(sparse lists, their postings, bytes of the sparse tier)"""
        n, p, b = C.c_uint32(), C.c_uint64(), C.c_uint64()
        N.check(N.lib().ss_bm25_sparse_info(self._h, C.byref(n), C.byref(p), C.byref(b)), "ss_bm25_sparse_info")
        return int(n.value), int(p.value), int(b.value)

    def set_coalescing(self, max_lexical_batch=1024, max_vector_batch=N.SS_VEC_BATCH, max_wait_us=0):
        """concurrent small host-pointer searches of this shard are merged into device batches behind the C ABI
        (ss_shard_set_coalescing; on by default, a batch size of 0 switches a kind off)"""
        N.check(N.lib().ss_shard_set_coalescing(self._h, int(max_lexical_batch), int(max_vector_batch), int(max_wait_us)),
                "ss_shard_set_coalescing")

    def generic_batches(self):
        """sub-batches the generic galloping kernels answered (ss_bm25_shape_stats): shapes beyond the specialised kernels"""
        v = C.c_uint64()
        N.check(N.lib().ss_bm25_shape_stats(self._h, C.byref(v)), "ss_bm25_shape_stats")
        return int(v.value)

    def one_launch_batches(self):
        """lexical host-pointer batches that took the one-launch path of small batches (ss_bm25_path_stats)"""
        v = C.c_uint64(0)
        N.check(N.lib().ss_bm25_path_stats(self._h, C.byref(v)), "ss_bm25_path_stats")
        return int(v.value)

    def coalescing_stats(self):
        """(lexical batches, lexical queries, vector batches, vector queries) served through the coalescer so far"""
        import ctypes as C
        v = [C.c_uint64() for _ in range(4)]
        N.check(N.lib().ss_shard_coalescing_stats(self._h, *[C.byref(x) for x in v]), "ss_shard_coalescing_stats")
        return tuple(int(x.value) for x in v)

    def profile(self, on=True):
        N.check(N.lib().ss_profile_enable(self._h, 1 if on else 0), "ss_profile_enable")

    def profile_read(self, kernel, reset=True):
        n, ms = C.c_uint64(), C.c_double()
        N.check(N.lib().ss_profile_read(self._h, kernel, C.byref(n), C.byref(ms), 1 if reset else 0), "ss_profile_read")
        return n.value, ms.value


def merge_results(mode, lex=None, vec=None, offset=0, length=10):
    """search.rs:1875-2119 on GLOBAL ids: concat over shards, RRF for Hybrid, sort desc, offset, truncate."""
    ld = np.ascontiguousarray(lex[0] if lex is not None else [], np.uint64)
    ls = np.ascontiguousarray(lex[1] if lex is not None else [], np.float32)
    vd = np.ascontiguousarray(vec[0] if vec is not None else [], np.uint64)
    vs = np.ascontiguousarray(vec[1] if vec is not None else [], np.float32)
    L = max(int(length), 1)
    od = np.empty(L, np.uint64)
    os_ = np.empty(L, np.float32)
    src = np.empty(L, np.uint8)
    n = N.check(N.lib().ss_merge_results(int(mode), N.ptr(ld, N.u64p), N.ptr(ls, N.f32p), len(ld), N.ptr(vd, N.u64p),
                                         N.ptr(vs, N.f32p), len(vd), int(offset), int(length), N.ptr(od, N.u64p),
                                         N.ptr(os_, N.f32p), N.ptr(src, N.u8p)), "ss_merge_results")
    return od[:n].copy(), os_[:n].copy(), src[:n].copy()


def rrf_merge_device(lex_doc, lex_count, vec_doc, vec_count, offset, length, stream, device=0):
    """ss_rrf_merge_dev on torch tensors already on `device`: [nq][k_lex] / [nq][k_vec] doc ids (int32 shard-local or int64
    global) + counts -> (doc int64 [nq][length], fused score, source uint8, count int32).  Either list may be None."""
    import torch
    ref = lex_doc if lex_doc is not None else vec_doc
    nq = ref.shape[0]
    wide = ref.dtype == torch.int64
    dev = ref.device
    od = torch.empty((nq, length), dtype=torch.int64, device=dev)
    os_ = torch.empty((nq, length), dtype=torch.float32, device=dev)
    src = torch.empty((nq, length), dtype=torch.uint8, device=dev)
    cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    kl = 0 if lex_doc is None else lex_doc.shape[1]
    kv = 0 if vec_doc is None else vec_doc.shape[1]
    N.check(N.lib().ss_rrf_merge_dev(device, nq, kl, None if lex_doc is None else lex_doc.data_ptr(),
                                     None if lex_doc is None else lex_count.data_ptr(), kv,
                                     None if vec_doc is None else vec_doc.data_ptr(),
                                     None if vec_doc is None else vec_count.data_ptr(), 1 if wide else 0, int(offset), int(length),
                                     od.data_ptr(), os_.data_ptr(), src.data_ptr(), cnt.data_ptr(), stream), "ss_rrf_merge_dev")
    return od, os_, src, cnt


class Index:
    """In-process multi-shard index: doc g lives in shard g % S with local id g // S (index.rs:5284)."""

    def __init__(self, shards: Sequence[Shard]):
        self.shards = list(shards)

    @property
    def shard_number(self):
        return len(self.shards)

    def search(self, query_terms: Optional[Sequence[int]] = None, query_vector=None,
               query_type_default=QueryType.Union, search_mode=SearchMode.Lexical, offset=0, length=10,
               result_type=ResultType.TopkCount, similarity_threshold=None, normalize_query=True,
               strict=False, not_terms=(), field_filter=None, facet_filter=None, ann_mode=None, result_sort=None) -> ResultObject:
        """<IndexArc as Search>::search (search.rs:1134-1150): field_filter applies to both sides (lexical: several indexed
        fields; vector: records of the listed fields), facet_filter to the lexical side, ann_mode to the vector side.
        result_sort (SearchMode.Lexical; see Shard.search_lexical_sorted): every shard returns its best offset + length under
        the sort, the lists are merged under the same order across shards -- the facet values of the two docs, each read from its
        own shard, then the score (result_ordering_root, min_heap.rs:56-300; search.rs:2088)"""
        S = self.shard_number
        ro = ResultObject()
        if result_sort:
            if search_mode != SearchMode.Lexical or not query_terms:
                raise ValueError("result_sort applies to lexical searches")
            rows = []
            for sh in self.shards:
                q = sh.make_queries([list(query_terms)], query_type_default, [list(not_terms)],
                                    field_filter=field_filter if sh.lexical_field_count > 1 else None)
                d, sc, tot = sh.search_lexical_sorted(q, list(result_sort), offset + length, facet_filter)
                ro.result_count_total += tot
                keys = []
                for sf in result_sort:
                    if sf[1] == "point":
                        vals, ty = sh.facet_point_distances(d, sf[0], sf[3], "sortkey").view(np.uint64), "f64"
                    else:
                        vals, ty = sh.facet_values(d, sf[0], sf[1]), sf[1]
                    keys.append([Shard._facet_order_key(x, ty, sf[2]) for x in vals])
                for i in range(len(d)):
                    rows.append((tuple(-kk[i] for kk in keys), -float(sc[i]), int(d[i]) * S + sh.shard_id))
            rows.sort()
            if result_type != ResultType.Count:
                ro.results = [Result(g, -ns, ResultSource.Lexical) for _, ns, g in rows[offset:offset + length]]
            ro.result_count = len(ro.results)
            return ro
        want_lex = search_mode in (SearchMode.Lexical, SearchMode.Hybrid) and query_terms
        want_vec = search_mode in (SearchMode.Vector, SearchMode.Hybrid) and query_vector is not None
        if want_vec and normalize_query:
            query_vector = normalize_f32(query_vector)  # search.rs:1464-1475 (Cosine, external inference)
        lex_d, lex_s, vec_d, vec_s = [], [], [], []
        for sh in self.shards:  # search.rs:1637-1743: each shard asked for (offset 0, length offset+length)
            lt = vt = 0
            if want_lex:
                r = sh.search_lexical_shard(query_terms, query_type_default, 0, offset + length, result_type, strict,
                                            not_terms, field_filter if sh.lexical_field_count > 1 else None, facet_filter)
                lex_d += [x.doc_id * S + sh.shard_id for x in r.results]  # search.rs:1671
                lex_s += [x.score for x in r.results]
                lt = r.result_count_total
            if want_vec:
                r = sh.search_vector_shard(query_vector, offset + length, similarity_threshold, strict, ann_mode, field_filter)
                vec_d += [x.doc_id * S + sh.shard_id for x in r.results]  # search.rs:1693
                vec_s += [x.score for x in r.results]
                vt = r.result_count_total
                ro.observed_vector_count += r.observed_vector_count  # search.rs:1897, 1923
                ro.observed_cluster_count += r.observed_cluster_count
            # search.rs:1884,1899 sum; Hybrid: max(lexical, vector) per shard (search.rs:1919-1921)
            ro.result_count_total += max(lt, vt) if search_mode == SearchMode.Hybrid else (lt + vt)
        if result_type != ResultType.Count:
            d, s, src = merge_results(search_mode, (lex_d, lex_s), (vec_d, vec_s), offset, length)
            ro.results = [Result(int(a), float(b), ResultSource(int(c))) for a, b, c in zip(d, s, src)]
        ro.result_count = len(ro.results)
        return ro
