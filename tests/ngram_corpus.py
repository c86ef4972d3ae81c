"""A corpus as the reference's DEFAULT indexing lays it out (NgramFF | NgramFFF, index.rs:1422-1424): four single terms with
positions and, for two sequences of them, the n-gram keys -- every doc that holds the sequence consecutively, the key's positions =
the places of its first word (tokenizer.rs:699), per posting the tf of each component term (index_posting.rs:666-722).
Shared by the GPU phrase test and the oracle's CPU test."""
import numpy as np

AB, ABC = (0, 1), (0, 1, 2)
KEYS = {AB: 0x7000_0000_0000 | 1, ABC: 0x7100_0000_0000 | 4}   # NgramType::NgramFF = 1, NgramFFF = 4 (index.rs:1854-1872)
# phrases as the query tokenizer resolves them -- entries = single terms or keys -- and the phrases over single terms that match
# the same docs
PHRASES = [[AB, 3], [3, AB], [ABC, 3], [2, AB], [AB, AB], [AB, 2], [3, ABC], [AB, 0]]
SAME_DOCS_AS = [[0, 1, 3], [3, 0, 1], [0, 1, 2, 3], [2, 0, 1], [0, 1, 0, 1], [0, 1, 2], [3, 0, 1, 2], [0, 1, 0]]


def KEY(t):
    return 1000 * (t + 1) * 8  # key_hash & 7 == 0: SingleTerm


class Corpus:
    pass


def build(O, corpus_fn, n_docs, dfs, plant, seed):
    C = Corpus()
    C.n_docs = n_docs
    C.dl, C.offs, C.docs, C.tfs, C.positions = corpus_fn(O, n_docs, dfs, seed, plant)
    per_term, at = [], 0   # term -> {doc: positions}
    for t in range(len(dfs)):
        m = {}
        for i in range(int(C.offs[t]), int(C.offs[t + 1])):
            m[int(C.docs[i])] = C.positions[at:at + int(C.tfs[i])].tolist()
            at += int(C.tfs[i])
        per_term.append(m)
    C.per_term = per_term

    def ngram(words):
        cand = set(per_term[words[0]])
        for w in words[1:]:
            cand &= set(per_term[w])
        rows = []
        for d in sorted(cand):
            sets = [set(per_term[w][d]) for w in words]
            ps = [p for p in per_term[words[0]][d] if all(p + i in sets[i] for i in range(1, len(words)))]
            if ps:
                rows.append((d, ps, [len(per_term[w][d]) for w in words]))
        return rows
    lut = lambda df: int(O.lib().so_int_to_byte4(int(df)))
    C.rows_of, C.ngram_terms = {}, []
    for words, key in KEYS.items():
        rows = ngram(list(words))
        C.rows_of[words] = rows
        C.ngram_terms.append((key, np.array([r[0] for r in rows], np.int64), np.array([len(r[1]) for r in rows], np.int64),
                              np.array([r[2] for r in rows], np.int64), [lut(len(per_term[w])) for w in words], [r[1] for r in rows]))
    C.terms = []
    for t in range(len(dfs)):
        ds = sorted(per_term[t])
        C.terms.append((KEY(t), np.array(ds, np.int64), np.array([len(per_term[t][d]) for d in ds], np.int64), [per_term[t][d] for d in ds]))
    # the oracle's lists: the single terms' + one per component of every key (same docs, the component's tf; the key's positions
    # behind its first component)
    o_offs, o_docs, o_tfs, o_pos, o_cnt, o_id = [0], [], [], [], [], {}
    for t in range(len(dfs)):
        o_id[("t", t)] = len(o_offs) - 1
        for d in sorted(per_term[t]):
            o_docs.append(d); o_tfs.append(len(per_term[t][d])); o_cnt.append(len(per_term[t][d])); o_pos += per_term[t][d]
        o_offs.append(len(o_docs))
    for words in KEYS:
        for c in range(len(words)):
            o_id[(words, c)] = len(o_offs) - 1
            for d, ps, ctf in C.rows_of[words]:
                o_docs.append(d); o_tfs.append(ctf[c]); o_cnt.append(len(ps) if c == 0 else 0)
                if c == 0:
                    o_pos += ps
            o_offs.append(len(o_docs))
    C.o_id = o_id
    C._o = (np.asarray(o_offs, np.uint64), np.asarray(o_docs, np.uint32), np.asarray(o_tfs, np.uint16), np.asarray(o_pos, np.uint16),
            np.asarray(o_cnt, np.uint16))

    def oracle_shard(O_):
        osh = O_.Shard(n_docs, C.dl, C._o[0], C._o[1], C._o[2])
        osh.set_positions(C._o[3], C._o[4])
        return osh
    C.oracle_shard = oracle_shard

    def oracle_query(ph, idf_component, idf_single):
        """entries of a phrase -> (unique lists, seq, places, idf) for so_search_phrase_items"""
        uniq, seq, places, idf, at_place = [], [], [], [], 0
        for e in ph:
            lists = [o_id[(e, c)] for c in range(len(e))] if isinstance(e, tuple) else [o_id[("t", e)]]
            for c, l in enumerate(lists):
                if l not in uniq:
                    uniq.append(l)
                    idf.append(idf_component(e, c) if isinstance(e, tuple) else idf_single(l))
            seq.append(uniq.index(lists[0]))
            places.append(at_place)
            at_place += len(lists)
        return uniq, seq, places, idf
    C.oracle_query = oracle_query
    return C
