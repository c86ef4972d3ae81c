"""ctypes front-end of oracle/ss_textindex.c -- TEST INFRASTRUCTURE ONLY.

A text-shaped corpus (Zipf tokens, topic clusters, real positions) indexed the way the reference indexes one field with its default
ngram_indexing (NgramFF | NgramFFF), and written as index.bin -- at sizes the pure-Python writer (ref_format.py) cannot reach."""
import ctypes as C

import numpy as np

from . import oracle as O

u8p, u16p, u32p, u64p = O.u8p, O.u16p, O.u32p, O.u64p
NGRAM_FF, NGRAM_FFF = 1, 8  # NgramSet bits (index.rs:1840-1850)


def _lib():
    L = O.lib()
    if not getattr(L, "_text_ready", False):
        L.so_text_build.restype = C.c_void_p
        L.so_text_build.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_double, C.c_double]
        L.so_text_free.argtypes = [C.c_void_p]
        L.so_text_info.argtypes = [C.c_void_p, u64p, u32p, u32p, u64p, u32p]
        L.so_text_doclen.restype = C.POINTER(C.c_uint8)
        L.so_text_doclen.argtypes = [C.c_void_p]
        L.so_text_doc_tokens.restype = C.c_uint32
        L.so_text_doc_tokens.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, u32p]
        L.so_text_ngram_key.restype = C.c_uint32
        L.so_text_ngram_key.argtypes = [C.c_void_p, C.c_uint32, u32p]
        L.so_text_key_hash.restype = C.c_uint64
        L.so_text_key_hash.argtypes = [C.c_void_p, C.c_uint32]
        L.so_text_key_df.restype = C.c_uint64
        L.so_text_key_df.argtypes = [C.c_void_p, C.c_uint32]
        L.so_text_key_postings.restype = C.c_uint64
        L.so_text_key_postings.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, u32p, u16p, u16p, u16p, C.c_uint64, u64p]
        L.so_text_write_index_bin.restype = C.c_int
        L.so_text_write_index_bin.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.POINTER(C.c_uint8)), u64p]
        L.so_text_free_bytes.argtypes = [C.POINTER(C.c_uint8)]
        L.so_text_build_fields.restype = C.c_void_p
        L.so_text_build_fields.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_double, C.c_double, C.c_uint32, C.c_uint32]
        L.so_text_doclen_fields.restype = C.POINTER(C.c_uint8)
        L.so_text_doclen_fields.argtypes = [C.c_void_p]
        L.so_text_doc_field_tokens.restype = C.c_uint32
        L.so_text_doc_field_tokens.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, u32p]
        L.so_text_key_entries.restype = C.c_uint64
        L.so_text_key_entries.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, u32p, u8p, u16p, u16p, u16p, C.c_uint64, u64p]
        L._text_ready = True
    return L


class TextCorpus:
    def __init__(self, seed, n_docs, vocab, n_frequent=64, ngrams=NGRAM_FF | NGRAM_FFF, topic_share=0.35, mean_len=100.0, n_fields=1,
                 longest_field=0):
        """n_fields > 1: the docs' tokens cut into that many consecutive spans indexed as separate fields (positions restart per field,
        n-grams stay inside one); longest_field = the longest_field_id the file declares"""
        self.n_docs, self.vocab, self.n_frequent, self.ngrams = int(n_docs), int(vocab), int(n_frequent), int(ngrams)
        self.n_fields, self.longest_field = int(n_fields), int(longest_field)
        self._h = _lib().so_text_build_fields(int(seed), int(n_docs), int(vocab), int(n_frequent), int(ngrams), float(topic_share), float(mean_len),
                                              int(n_fields), int(longest_field))
        if not self._h:
            raise ValueError("so_text_build refused the arguments")
        nt, np_ = C.c_uint64(), C.c_uint64()
        nk, ne, ng = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _lib().so_text_info(self._h, C.byref(nt), C.byref(nk), C.byref(ne), C.byref(np_), C.byref(ng))
        self.n_tokens, self.n_keys, self.n_keys_nonempty, self.n_postings, self.n_ngram_keys = nt.value, nk.value, ne.value, np_.value, ng.value
        self.doclen = np.ctypeslib.as_array(_lib().so_text_doclen(self._h), shape=(self.n_docs,)).copy()
        # [n_fields][n_docs] length bytes of the indexed fields (one field: the same bytes)
        self.doclen_fields = np.ctypeslib.as_array(_lib().so_text_doclen_fields(self._h), shape=(self.n_fields, self.n_docs)).copy()

    def close(self):
        if getattr(self, "_h", None):
            _lib().so_text_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def doc_tokens(self, d):
        out = np.zeros(1500, np.uint32)
        n = _lib().so_text_doc_tokens(self._h, int(d), len(out), O._p(out, u32p))
        return out[:n].copy()

    def doc_field_tokens(self, d, f):
        out = np.zeros(1500, np.uint32)
        n = _lib().so_text_doc_field_tokens(self._h, int(d), int(f), len(out), O._p(out, u32p))
        return out[:n].copy()

    def key_entries(self, key, component=0, positions=True):
        """several indexed fields: (docs, fields, tfs, counts, positions) of the key's (doc, field) entries -- for component c of an n-gram
        key the fields in which the component TERM stands, counts / positions = the key's own (component 0 only)"""
        n = _lib().so_text_key_entries(self._h, int(key), int(component), None, None, None, None, None, 0, None)
        docs, flds, tfs, cnt = np.zeros(n, np.uint32), np.zeros(n, np.uint8), np.zeros(n, np.uint16), np.zeros(n, np.uint16)
        npos = C.c_uint64()
        _lib().so_text_key_entries(self._h, int(key), int(component), O._p(docs, u32p), O._p(flds, u8p), O._p(tfs, u16p), O._p(cnt, u16p), None, 0,
                                   C.byref(npos))
        pos = np.zeros(npos.value if positions else 0, np.uint16)
        if positions and npos.value:
            _lib().so_text_key_entries(self._h, int(key), int(component), None, None, None, None, O._p(pos, u16p), len(pos), C.byref(npos))
        return docs, flds, tfs, cnt, pos

    def ngram_key(self, ranks):
        """key id of the n-gram over these 2 / 3 ranks, None if the corpus holds none"""
        c = np.ascontiguousarray(list(ranks) + [0] * (3 - len(ranks)), np.uint32)
        k = _lib().so_text_ngram_key(self._h, len(ranks), O._p(c, u32p))
        return None if k == 0xFFFFFFFF else int(k)

    def key_hash(self, key):
        return int(_lib().so_text_key_hash(self._h, int(key)))

    def key_df(self, key):
        return int(_lib().so_text_key_df(self._h, int(key)))

    def key_postings(self, key, component=0, positions=True):
        """(docs, tfs, counts, positions): tfs = the positions count of a single term / of component `component` of an n-gram key,
        counts = the key's own positions per posting"""
        n = self.key_df(key)
        docs, tfs, cnt = np.zeros(n, np.uint32), np.zeros(n, np.uint16), np.zeros(n, np.uint16)
        npos = C.c_uint64()
        _lib().so_text_key_postings(self._h, int(key), int(component), O._p(docs, u32p), O._p(tfs, u16p), O._p(cnt, u16p), None, 0, C.byref(npos))
        pos = np.zeros(npos.value if positions else 0, np.uint16)
        if positions and npos.value:
            _lib().so_text_key_postings(self._h, int(key), int(component), None, None, None, O._p(pos, u16p), len(pos), C.byref(npos))
        return docs, tfs, cnt, pos

    def write_index_bin(self, segment_number_bits=11, key_head_size=23, positions_limit=32768):
        out = C.POINTER(C.c_uint8)()
        n = C.c_uint64()
        if _lib().so_text_write_index_bin(self._h, segment_number_bits, key_head_size, positions_limit, C.byref(out), C.byref(n)) != 0:
            raise ValueError("so_text_write_index_bin")
        data = bytes(np.ctypeslib.as_array(out, shape=(n.value,)))
        _lib().so_text_free_bytes(out)
        return data

    def query_entries(self, ranks):
        """a quoted phrase as the query tokenizer resolves it (tokenizer.rs:900-1370, every word with QueryType::Phrase): greedily a
        trigram key over three frequent words, else a bigram key over two, else the word -> list of (key id, component ranks)
        (an n-gram key the corpus does not hold resolves to key id None: the reference then finds no posting list)"""
        out, i, n = [], 0, len(ranks)
        fr = [r < self.n_frequent for r in ranks]
        while i < n:
            if i + 2 < n and (self.ngrams & NGRAM_FFF) and fr[i] and fr[i + 1] and fr[i + 2]:
                out.append((self.ngram_key(ranks[i:i + 3]), tuple(int(r) for r in ranks[i:i + 3]))); i += 3
            elif i + 1 < n and (self.ngrams & NGRAM_FF) and fr[i] and fr[i + 1]:
                out.append((self.ngram_key(ranks[i:i + 2]), tuple(int(r) for r in ranks[i:i + 2]))); i += 2
            else:
                out.append((int(ranks[i]), (int(ranks[i]),))); i += 1
        return out
