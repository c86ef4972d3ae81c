"""A WHOLE index.bin assembled BY HAND -- version header, two levels (the second incomplete), the per-level header, the segment
head table, sorted key heads of 20 / 22 / 23 bytes and the key bodies -- from the reference's writers, independently of
oracle/ref_format.py (the restated writer the other index.bin fixtures come from): the FILE WALK of ss_index_bin_open
(seekstorm_amd/csrc/ref_format.hip; the reference's reader is index.rs:3263-3740) is pinned by bytes that cite the line that
writes each of them (paths relative to /root/reference/seekstorm/src).  The key bodies are those of hand_assembled_blocks.py
(H1 .. H4, every byte cited there), re-based the way commit_segment stacks the bodies of a segment behind one another.

File layout (one shard, one indexed field):

  [u16 INDEX_FORMAT_VERSION_MAJOR = 6][u16 MINOR = 1]                       index.rs:103-107 (INDEX_HEADER_SIZE 4), written 2349-2363
  per level (= 65 536 docs, commit.rs commit_level 232-372):
    level 0 only: [u16 longest_field_id]                                    commit.rs:264-275 (committed_doc_count / 65536 == 0)
    per indexed field: [65 536 doc-length bytes]                            commit.rs:277-282; zeroed again after the level, 341-343
    [u64 indexed_doc_count, cumulative][u64 positions_sum_normalized, cumulative]   commit.rs:298-311; read back index.rs:3419-3431
    [segment_number1 x (u32 block_length, u32 key_count)]                   table skipped first (313-316), filled by commit_segment
                                                                            (key_count 469-479, block_length 529-537), written 352-359
    per segment: [key_count x key head, ascending key_hash][key bodies]     commit_segment 467-552: heads first (481-483), keys sorted
                                                                            (485-486), block_length = heads + bodies (527)
  key head (compress_postinglist.rs:339-409): u64 key_hash | u16 posting_count - 1 | u16 max_docid | u16 max_p_docid |
    [22-byte heads: 2, 23-byte heads: 3 bytes posting_count_ngram_i_compressed] | u16 pointer_pivot_p_docid | u32 compression_type_pointer
  compression_type_pointer = key_body_offset | CompressionType << 30 (compress_postinglist.rs:722, 788, 867) with key_body_offset =
    key_rank_position_pointer_w - key_body_pointer_wstart (commit.rs:507-508): the offset of the key's rank/position pointers (R)
    inside the SEGMENT's key-body slice = the bodies of the keys before it + its own position records.
  The segment of a key is hash32(term) & mask in the reference (tokenizer.rs:660) -- another hash of the same term; a reader
  walks every segment and never relies on which one a key sits in.
"""
import importlib.util
import os

_spec = importlib.util.spec_from_file_location("hand_assembled_blocks", os.path.join(os.path.dirname(os.path.abspath(__file__)), "hand_assembled_blocks.py"))
B = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(B)

SEGMENT_NUMBER_BITS = 1          # 2 segments (the reference default is 11; the reader takes it from index.json)
N_SEG = 1 << SEGMENT_NUMBER_BITS
LEVEL1_DOCS = 50_000             # level 1 is incomplete
N_DOCS = 65_536 + LEVEL1_DOCS


def _u16(v): return list(int(v).to_bytes(2, "little"))
def _u32(v): return list(int(v).to_bytes(4, "little"))
def _u64(v): return list(int(v).to_bytes(8, "little"))


def byte4_to_int(b):
    """DOCUMENT_LENGTH_COMPRESSION[b] (index.rs:4255-4279): the value a length byte stands for"""
    if b < 24:
        return b
    i = b - 24
    bits, shift = i & 7, i >> 3
    return 24 + bits if shift == 0 else 24 + ((bits | 8) << (shift - 1))


# key hashes: low 3 bits 0 = NgramType::SingleTerm (index.rs:1853-1872)
KEY_A, KEY_B, KEY_C = 0x1111_2222_3333_4440, 0x5555_0000_0000_0008, 0x9999_AAAA_BBBB_CCC0

# the parts of the hand-assembled bodies WITHOUT their "bytes of other keys" prefix: (records, pointers, container)
PARTS = {
    "H1": (B.H1_RECORD_P3 + B.H1_RECORD_P2, B.H1_POINTERS, B.H1_DOCIDS, B.H1),
    "H2": (B.H2_RECORD_P3, B.H2_POINTERS, B.H2_DOCIDS, B.H2),
    "H3": ([], B.H3_POINTERS, list(B.H3["body"][len(B.H3_PREFIX) + len(B.H3_POINTERS):]), B.H3),
    "H4": (B.H4_REC_P3 + B.H4_REC_P2 + B.H4_REC_P1, B.H4_POINTERS, B.H4_CONTAINER, B.H4),
}
# level -> segment -> [(key_hash, body name)] in any order (the writer sorts them)
LAYOUT = {0: {0: [(KEY_B, "H3"), (KEY_A, "H1")], 1: [(KEY_C, "H4")]},
          1: {0: [(KEY_A, "H2")], 1: []}}


def doclen_bytes(level):
    """the level's 65 536 length bytes: any byte is a legal compressed length"""
    if level == 0:
        return [8 + (d % 50) for d in range(65536)]
    return [20] * LEVEL1_DOCS + [0] * (65536 - LEVEL1_DOCS)  # beyond the indexed docs the array still holds its reset value 0 (commit.rs:341-343)


def segment_bytes(entries, key_head_size):
    """commit_segment (commit.rs:467-552) for one segment of one level"""
    entries = sorted(entries, key=lambda e: e[0])          # key_list.sort_unstable(): 485-486
    heads, bodies = [], []
    for key, name in entries:
        records, pointers, container, blk = PARTS[name]
        r = len(bodies) + len(records)                     # key_body_offset: 497-498 + 507-508
        ctype = blk["compression_type_pointer"] >> 30      # Array 1 / Bitmap 2 / Rle 3 (index.rs:838-843)
        h = _u64(key)                                      # compress_postinglist.rs:339-343
        h += _u16(blk["posting_count"] - 1)                # 345-349
        h += _u16(0)                                       # max_docid 351-355: the block-max posting; this reader derives its own bounds
        h += _u16(0)                                       # max_p_docid 357-361
        h += [0] * (key_head_size - 20)                    # 363-395: posting_count_ngram_i_compressed, 0 for a SingleTerm key
        h += _u16(blk["pointer_pivot_p_docid"])            # 397-401
        h += _u32(r | (ctype << 30))                       # 403-407; value: 722 / 788 / 867
        assert len(h) == key_head_size
        heads += h
        bodies += records + pointers + container           # [records, stacked down from R][pointers up from R][container]
    return heads + bodies, len(entries)


def build(key_head_size=20):
    assert key_head_size in (20, 22, 23)
    out = _u16(6) + _u16(1)                                # index.rs:105, 107, 2349-2363
    docs_cum, psum_cum = 0, 0
    for level in (0, 1):
        if level == 0:
            out += _u16(0)                                 # longest_field_id: commit.rs:264-275
        dl = doclen_bytes(level)
        out += dl                                          # commit.rs:277-282
        n_level = 65536 if level == 0 else LEVEL1_DOCS
        docs_cum += n_level
        psum_cum += sum(byte4_to_int(b) for b in dl[:n_level])
        out += _u64(docs_cum) + _u64(psum_cum)             # commit.rs:298-311
        table, payload = [], []
        for seg in range(N_SEG):
            sb, n_keys = segment_bytes(LAYOUT[level][seg], key_head_size)
            table += _u32(len(sb)) + _u32(n_keys)          # (block_length, key_count): commit.rs:529-537, 469-479; read index.rs:3433-3446
            payload += sb
        out += table + payload                             # the table sits in front of the segments: 313-316, 352-359
    return bytes(out), psum_cum


# what a reader must find: term id = rank of the key hash; postings with shard-local doc ids (level << 16 | doc)
EXPECT = {
    KEY_A: ([d for d in B.H1["docs"]] + [65536 + d for d in B.H2["docs"]], B.H1["tfs"] + B.H2["tfs"]),
    KEY_B: (B.H3["docs"], B.H3["tfs"]),
    KEY_C: (B.H4["docs"], B.H4["tfs"]),
}
