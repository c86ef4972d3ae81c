"""Key bodies of the reference's block format assembled BY HAND, byte by byte, from the reference's WRITERS -- independently of
oracle/ref_format.py (the restated writer the other fixtures come from), so that a misreading shared by that restatement and
the decoder (seekstorm_amd/csrc/ref_format.hip) cannot hide.  Every byte cites the line of the writer that produces it
(paths relative to /root/reference/seekstorm/src).  tests/test_ref_format.py::test_hand_assembled_key_bodies feeds them to
ss_ref_decode_block.

Layout of one key's bytes inside a 65 536-doc block (compress_postinglist.rs:832-946, the Rle writer; Array 694- and Bitmap
759- lay the first two regions out the same way):

    [ position records, stacked DOWNWARD from R ]  [ rank/position pointers, upward from R ]  [ doc-id container ]
      docid_iterator:  key_position_pointer_w -= size (compress_postinglist.rs:475, 505)
      R = rank_position_pointer_range = low 30 bits of compression_type_pointer; bits 30..31 = CompressionType
          (index.rs:838-843: Array = 1, Bitmap = 2, Rle = 3)

  pointer of posting p: 2 bytes while p < pointer_pivot_p_docid, 3 bytes from there on (single.rs:77-83);
    not embedded : the running sum of record sizes, low byte first, top bit clear (compress_postinglist.rs:477-482, 507-515);
                   the record lies at R - that sum (index.rs:2987-2996)
    embedded     : the positions themselves, top bit set (index_posting.rs:605-660)
  record of a posting that is not embedded, one indexed field, SingleTerm key (index_posting.rs write_field_vec 848-872, then
    compress_positions compress_postinglist.rs:949-977): positions_count as a VINT whose LAST byte carries STOP_BIT 0x80, then
    every position (delta form) as a VINT of 1-3 bytes, last byte | 0x80.
  tf of a posting = positions_count (add_result.rs:2036-2197): the count bits of an embedded pointer, else the record's first VINT.
"""

STOP = 0x80


def _u16(v):
    return [v & 0xFF, v >> 8]


# ---------------------------------------------------------------------------------------------------------------- H1
# CompressionType::Array, four postings, every pointer 2 bytes: pointer_pivot_p_docid = 4 (index_posting.rs:193-196: set to
# posting_count + 1 before each 2-byte posting is added, so after n such postings it reads n).
#   p0 doc 3      1 position  {5}            embedded   (index_posting.rs:449-451: 1 position of <= 14 bits)
#   p1 doc 17     2 positions {3, 9}         embedded   (452-454: 2 positions of <= 7 bits each)
#   p2 doc 300    3 positions {10, 0, 188}   record     (a 2-byte pointer embeds at most 2 positions: 449)
#   p3 doc 65535  130 positions, all 0       record, positions_count needs a 2-byte VINT
H1_PREFIX = [0xAA, 0xBB, 0xCC, 0xDD, 0xEE, 0xFF, 0x11]  # 7 bytes of other keys in front of this key's records
H1_RECORD_P3 = (
    [130 >> 7, (130 & 0x7F) | STOP]                    # positions_count 130: write_field_vec 861-865 (>= 128: high 7 bits, then low | STOP)
    + [0 | STOP] * 130                                 # 130 positions of delta 0: compress_positions 955-957
)                                                      # 132 bytes
H1_RECORD_P2 = [
    3 | STOP,                                          # positions_count 3: write_field_vec 858-860 (< 128: one byte | STOP)
    10 | STOP,                                         # position 10: compress_positions 955-957
    0 | STOP,                                          # delta 0
    188 >> 7, (188 & 0x7F) | STOP,                     # delta 188: compress_positions 958-965 (two bytes, STOP on the last)
]                                                      # 5 bytes
H1_R = len(H1_PREFIX) + len(H1_RECORD_P3) + len(H1_RECORD_P2)  # = 144: p2's record ends AT R, p3's lies below it (stack)
H1_POINTERS = [
    # p0: data = 5 in 14 bits (index_posting.rs:605-619: remaining_bits = 2*8 - 0 - 2 = 14, position_bits = 14 / 1);
    #     byte0 = data & 0xFF, byte1 = data >> 8 | 0b1000_0000 | (positions - 1) << 6   (622-625)
    5 & 0xFF, (5 >> 8) | 0x80 | (0 << 6),
    # p1: data = 3 << 7 | 9 (position_bits 14 / 2 = 7, then 7 / 1 = 7; the first position ends up in the high bits: 615-618)
    ((3 << 7) | 9) & 0xFF, (((3 << 7) | 9) >> 8) | 0x80 | (1 << 6),
    # p2: running record size 5 (compress_postinglist.rs:474, 477-482), top bit clear
    5 & 0xFF, (5 >> 8) & 0x7F,
    # p3: running record size 5 + 132 = 137
    137 & 0xFF, (137 >> 8) & 0x7F,
]
H1_DOCIDS = _u16(3) + _u16(17) + _u16(300) + _u16(65535)  # Array container: u16 little endian, ascending (single.rs:176-183)
H1 = dict(
    name="H1 array, 2-byte pointers, embedded and recorded postings",
    block_id=5, compression_type_pointer=(1 << 30) | H1_R, posting_count=4, pointer_pivot_p_docid=4,
    body=bytes(H1_PREFIX + H1_RECORD_P3 + H1_RECORD_P2 + H1_POINTERS + H1_DOCIDS),
    docs=[3, 17, 300, 65535], tfs=[1, 2, 3, 130],
    # absolute positions: the stored values are the first position, then gap - 1 (index_posting.rs:55-64; the reader adds
    # value + 1, get_next_position_singlefield in add_result.rs:3596-3684)
    positions=[5] + [3, 13] + [10, 11, 200] + list(range(130)))

# ---------------------------------------------------------------------------------------------------------------- H2
# CompressionType::Array, pivot INSIDE the list: p0 has a 2-byte pointer, p1..p3 3-byte pointers (offset of pointer p >= pivot:
# 3 p - pivot, index.rs:2978).  The writer moves the pivot when 32 768 bytes of records are reached (index_posting.rs:579-587);
# the reader only ever looks at the stored value, so a small pivot is a legal header and keeps the fixture readable.
#   p0 doc 0      1 position  {300}              embedded, 2 bytes
#   p1 doc 9      4 positions {1, 2, 3, 40}      embedded, 3 bytes (index_posting.rs:466-470: <= 5, 5, 5, 6 bits)
#   p2 doc 10     2 positions {1000, 2000}       embedded, 3 bytes (459-461: <= 10 and <= 11 bits)
#   p3 doc 40000  5 positions, all 0             record
H2_PREFIX = [0x01, 0x02, 0x03]
H2_RECORD_P3 = [5 | STOP] + [0 | STOP] * 5             # 6 bytes
H2_R = len(H2_PREFIX) + len(H2_RECORD_P3)              # = 9
_d1 = (((((1 << 5) | 2) << 5) | 3) << 6) | 40          # remaining_bits 3*8 - 1 - 2 = 21: 21/4 = 5, 16/3 = 5, 11/2 = 5, 6/1 = 6 (605-619)
_d2 = (1000 << 11) | 2000                              # 21/2 = 10, then 11
H2_POINTERS = [
    300 & 0xFF, (300 >> 8) | 0x80 | (0 << 6),                        # p0: 2-byte embedded form (622-625)
    _d1 & 0xFF, (_d1 >> 8) & 0xFF, (_d1 >> 16) | 0x80 | (3 << 5),    # p1: 3-byte form, (positions - 1) << 5 (636-640)
    _d2 & 0xFF, (_d2 >> 8) & 0xFF, (_d2 >> 16) | 0x80 | (1 << 5),    # p2
    6 & 0xFF, (6 >> 8) & 0xFF, (6 >> 16) & 0x7F,                     # p3: running record size 6 (compress_postinglist.rs:507-515)
]
H2_DOCIDS = _u16(0) + _u16(9) + _u16(10) + _u16(40000)
H2 = dict(
    name="H2 array, pivot inside the list, 3-byte embedded forms",
    block_id=0, compression_type_pointer=(1 << 30) | H2_R, posting_count=4, pointer_pivot_p_docid=1,
    body=bytes(H2_PREFIX + H2_RECORD_P3 + H2_POINTERS + H2_DOCIDS),
    docs=[0, 9, 10, 40000], tfs=[1, 4, 2, 5],
    positions=[300] + [1, 4, 8, 49] + [1000, 3001] + [0, 1, 2, 3, 4])

# ---------------------------------------------------------------------------------------------------------------- H3
# CompressionType::Bitmap: 8 192 bytes, bit d & 7 of byte d >> 3 (single.rs:235-262 reads it as 1 024 u64 words and walks them
# with tzcnt: the same bits).  The writer chooses a bitmap for a dense block; the reader takes the type from the header.
# Five postings, each with one embedded position; no records, so R directly follows the prefix.
H3_DOCS = [0, 1, 63, 64, 65535]
H3_PREFIX = [0x7E]
H3_R = len(H3_PREFIX)
H3_POINTERS = []
for _pos in (0, 7, 8, 16383, 129):                     # one position each, <= 14 bits: embedded 2-byte pointers (449-451, 622-625)
    H3_POINTERS += [_pos & 0xFF, (_pos >> 8) | 0x80 | (0 << 6)]
_bm = bytearray(8192)
for _d in H3_DOCS:
    _bm[_d >> 3] |= 1 << (_d & 7)
H3 = dict(
    name="H3 bitmap",
    block_id=2, compression_type_pointer=(2 << 30) | H3_R, posting_count=5, pointer_pivot_p_docid=5,
    body=bytes(H3_PREFIX + H3_POINTERS) + bytes(_bm),
    docs=H3_DOCS, tfs=[1, 1, 1, 1, 1], positions=[0, 7, 8, 16383, 129])

# ---------------------------------------------------------------------------------------------------------------- H4
# CompressionType::Rle (compress_postinglist.rs:832-946): u16 runs_count, then per run u16 start and u16 length, where length
# counts the docs AFTER the start (a run of one doc has length 0: 895-897, 919-936).
#   runs: 10..12, 100, 65530..65535  -> 10 postings; tf = rank + 1 for the first four (records), 1 for the rest (embedded)
H4_DOCS = [10, 11, 12, 100, 65530, 65531, 65532, 65533, 65534, 65535]
H4_TFS = [1, 2, 3, 4, 1, 1, 1, 1, 1, 1]
H4_PREFIX = [0x55, 0x66]
# p1, p2, p3 are recorded (3 and 4 positions do not fit a 2-byte pointer; p1's two positions {200, 1} do not either: 200 > 7 bits)
H4_REC_P1 = [2 | STOP, 200 >> 7, (200 & 0x7F) | STOP, 1 | STOP]   # count 2, delta 200 (two bytes), delta 1 -> 4 bytes
H4_REC_P2 = [3 | STOP, 4 | STOP, 4 | STOP, 4 | STOP]              # 4 bytes
H4_REC_P3 = [4 | STOP, 1 | STOP, 1 | STOP, 1 | STOP, 1 | STOP]    # 5 bytes
H4_R = len(H4_PREFIX) + len(H4_REC_P3) + len(H4_REC_P2) + len(H4_REC_P1)  # = 15 (p1's record is the one touching R)
H4_POINTERS = (
    [9 & 0xFF, (9 >> 8) | 0x80 | (0 << 6)]             # p0: position 9 embedded
    + [4, 0]                                           # p1: running size 4
    + [8, 0]                                           # p2: 4 + 4
    + [13, 0]                                          # p3: 8 + 5
    + [1 & 0xFF, 0x80] * 6                             # p4..p9: position 1 embedded
)
H4_CONTAINER = _u16(3) + _u16(10) + _u16(2) + _u16(100) + _u16(0) + _u16(65530) + _u16(5)
H4 = dict(
    name="H4 rle",
    block_id=1, compression_type_pointer=(3 << 30) | H4_R, posting_count=10, pointer_pivot_p_docid=10,
    body=bytes(H4_PREFIX + H4_REC_P3 + H4_REC_P2 + H4_REC_P1 + H4_POINTERS + H4_CONTAINER),
    docs=H4_DOCS, tfs=H4_TFS, positions=[9] + [200, 202] + [4, 9, 14] + [1, 3, 5, 7] + [1] * 6)

# ---------------------------------------------------------------------------------------------------------------- H7
# Positions of 16 384 and more: compress_positions' THREE-byte form (compress_postinglist.rs:966-975) stores
#   delta >> 13 (& 0x7F), (delta >> 7) & 0x7F, (delta & 0x7F) | STOP
# -- bit 13 of delta lands in the first AND the second byte --, and the readers put it back together as
#   b0 << 13 | b1 << 7 | b2 & 0x7F      (get_next_position_singlefield / _multifield, add_result.rs:51-56, 82-87),
# which is NOT the three-byte form of the counts (write_field_vec 866-873: >> 14, >> 7).  One posting, doc 77, positions
# {20000, 60000, 60001}: stored 20000, 60000 - 20000 - 1 = 39999, 0.
H7_RECORD = [
    3 | STOP,                                                      # positions_count 3
    (20000 >> 13) & 0x7F, (20000 >> 7) & 0x7F, (20000 & 0x7F) | STOP,   # = 0x02, 0x1C, 0xA0
    (39999 >> 13) & 0x7F, (39999 >> 7) & 0x7F, (39999 & 0x7F) | STOP,   # = 0x04, 0x38, 0xBF  (39999 >> 7 = 312 = 0b1_0011_1000: bit 13 again)
    0 | STOP,
]                                                                  # 8 bytes
H7 = dict(
    name="H7 three-byte positions",
    block_id=3, compression_type_pointer=(1 << 30) | len(H7_RECORD), posting_count=1, pointer_pivot_p_docid=1,
    body=bytes(H7_RECORD + [8, 0] + _u16(77)),
    docs=[77], tfs=[3], positions=[20000, 60000, 60001])

# ================================================================================================ several indexed fields
# An index with SEVERAL indexed fields: a posting carries a field vector [(field id, positions count)] and the positions of
# every listed field, each field's positions restarting at an absolute first value (index_posting.rs:395-441 collect them per
# field; reader: decode_positions_multiterm_multifield add_result.rs:1485-2034, get_next_position_multifield, the phrase check
# 3248-3386).  Three indexed fields -> indexed_field_id_bits = 2 (index.rs:2569-2570); longest_field_id = 1.
#   embedded pointer (index_posting.rs:592-660): data = [field ids, 2 bits each, unless only the longest field occurs],
#   then the positions, position i of n taking floor(remaining_bits / (n - i)) bits (605-619), where remaining_bits =
#   pointer bits - (1 for a 3-byte pointer) - (3 for "only the longest field" | 4 + 2 per named field) (597-604).
# ---------------------------------------------------------------------------------------------------------------- H5
# 2-byte pointers (pointer_pivot_p_docid = posting count), CompressionType::Array
#   p0 doc 5      field 1 (the longest) only, positions {3, 20}   embedded: 16 - 3 = 13 bits -> 6 + 7; stored 3 and 20 - 3 - 1 = 16
#   p1 doc 9      field 2 only, position {100}                    embedded: 16 - (4 + 2) = 10 bits
#   p2 doc 700    fields 0 {7} and 2 {12}                         embedded: 16 - (4 + 4) = 8 bits -> 4 + 4
#   p3 doc 40000  fields 0 {1, 5} and 1 {0, 2, 9}                 record (5 positions: never embedded, index_posting.rs:437)
H5_RECORD_P3 = [
    # field vector, write_field_vec 897-925 (several fields, not only the longest): value = count << 2 | field id;
    # entry 0: (field 0, 2 positions): 2 << 2 | 0 = 8, meta bits 1 + 2 + 2 = 5 <= 6 -> one byte | STOP; not the last entry: no field stop bit
    8 | STOP,
    # entry 1: (field 1, 3 positions): 3 << 2 | 1 = 13, meta bits 0 + 2 + 2 = 4 -> one byte | STOP | FIELD_STOP_BIT_2 (0x40: last entry, i > 0)
    13 | STOP | 0x40,
    1 | STOP, 3 | STOP,            # field 0: position 1, then 5 - 1 - 1 = 3 (compress_positions 955-957)
    0 | STOP, 1 | STOP, 6 | STOP,  # field 1: position 0, 2 - 0 - 1 = 1, 9 - 2 - 1 = 6
]                                  # 7 bytes
H5_R = len(H5_RECORD_P3)
_h5_p0 = (3 << 7) | 16                                  # 6 + 7 bits
_h5_p1 = (2 << 10) | 100                                # field id 2, then 10 position bits
_h5_p2 = (((0 << 2) | 2) << 8) | (7 << 4) | 12          # field ids 0, 2; 4 + 4 position bits
H5_POINTERS = [
    _h5_p0 & 0xFF, (_h5_p0 >> 8) | 0xC0 | (1 << 5),     # only the longest field: 0b1100_0000 | (positions - 1) << 5 (626-628)
    _h5_p1 & 0xFF, (_h5_p1 >> 8) | 0x80 | (0 << 4),     # one named field: 0b1000_0000 | (positions - 1) << 4 (629-631)
    _h5_p2 & 0xFF, (_h5_p2 >> 8) | 0xB0,                # two named fields: 0b1011_0000 (632-633)
    7 & 0xFF, (7 >> 8) & 0x7F,                          # p3: running record size 7, top bit clear
]
H5 = dict(
    name="H5 three indexed fields, 2-byte pointers",
    n_fields=3, longest_field_id=1,
    block_id=2, compression_type_pointer=(1 << 30) | H5_R, posting_count=4, pointer_pivot_p_docid=4,
    body=bytes(H5_RECORD_P3 + H5_POINTERS + _u16(5) + _u16(9) + _u16(700) + _u16(40000)),
    docs=[5, 9, 700, 40000],
    entries=[[(1, [3, 20])], [(2, [100])], [(0, [7]), (2, [12])], [(0, [1, 5]), (1, [0, 2, 9])]])

# ---------------------------------------------------------------------------------------------------------------- H6
# 3-byte pointers from the first posting on (pointer_pivot_p_docid = 0: offset of pointer p = 3 p - 0), Array container
#   p0 doc 1      field 1 (the longest) only, positions {2, 5, 9, 40}     embedded: 24 - 1 - 3 = 20 bits -> 5 + 5 + 5 + 5; stored 2, 2, 3, 30
#   p1 doc 2      field 0 only, positions {1000, 1010}                    embedded: 24 - 1 - (4 + 2) = 17 bits -> 8 + 9 ... 1000 needs 10 bits:
#                                                                         NOT embeddable (index_posting.rs:525-530) -> record
#   p2 doc 3      field 0 only, positions {100, 400}                      embedded: 17 bits -> 8 + 9; stored 100, 299
#   p3 doc 50     fields 0 {3} and 2 {4, 6}                               embedded: 24 - 1 - (4 + 4) = 15 bits -> 5 + 5 + 5; tag (1, 2)
#   p4 doc 51     fields 0 {1}, 1 {2} and 2 {3}                           embedded: 24 - 1 - (4 + 6) = 13 bits -> 4 + 4 + 5; tag three fields
H6_RECORD_P1 = [
    # (field 0, 2 positions): 2 << 2 | 0 = 8; the only entry: i == 0 and last -> FIELD_STOP_BIT_1 (0x20); meta bits 1 + 2 + 2 = 5 -> one byte
    8 | STOP | 0x20,
    1000 >> 7, (1000 & 0x7F) | STOP,   # position 1000: two bytes (compress_positions 958-965)
    9 | STOP,                          # 1010 - 1000 - 1
]                                      # 4 bytes
H6_R = len(H6_RECORD_P1)
_h6_p0 = (2 << 15) | (2 << 10) | (3 << 5) | 30
_h6_p2 = (0 << 17) | (100 << 9) | 299
_h6_p3 = (((0 << 2) | 2) << 15) | (3 << 10) | (4 << 5) | 1          # field ids 0, 2; positions 3 | 4, 6 - 4 - 1 = 1
_h6_p4 = (((((0 << 2) | 1) << 2) | 2) << 13) | (1 << 9) | (2 << 5) | 3  # field ids 0, 1, 2; 4 + 4 + 5 bits
H6_POINTERS = [
    _h6_p0 & 0xFF, (_h6_p0 >> 8) & 0xFF, (_h6_p0 >> 16) | 0xC0 | (3 << 4),   # only the longest field: 0b1100_0000 | (positions - 1) << 4 (641-643)
    4 & 0xFF, (4 >> 8) & 0xFF, (4 >> 16) & 0x7F,                             # p1: running record size 4
    _h6_p2 & 0xFF, (_h6_p2 >> 8) & 0xFF, (_h6_p2 >> 16) | 0x80 | (1 << 3),   # one named field: 0b1000_0000 | (positions - 1) << 3 (645-648)
    _h6_p3 & 0xFF, (_h6_p3 >> 8) & 0xFF, (_h6_p3 >> 16) | 0x80 | 0x28,       # two named fields with 1 and 2 positions: 0b0010_1000 (653-654)
    _h6_p4 & 0xFF, (_h6_p4 >> 8) & 0xFF, (_h6_p4 >> 16) | 0x80 | 0x38,       # three named fields: 0b0011_1000 (649-650)
]
H6 = dict(
    name="H6 three indexed fields, 3-byte pointers",
    n_fields=3, longest_field_id=1,
    block_id=0, compression_type_pointer=(1 << 30) | H6_R, posting_count=5, pointer_pivot_p_docid=0,
    body=bytes(H6_RECORD_P1 + H6_POINTERS + _u16(1) + _u16(2) + _u16(3) + _u16(50) + _u16(51)),
    docs=[1, 2, 3, 50, 51],
    entries=[[(1, [2, 5, 9, 40])], [(0, [1000, 1010])], [(0, [100, 400])], [(0, [3]), (2, [4, 6])], [(0, [1]), (1, [2]), (2, [3])]])

# ================================================================================================ n-gram keys, one indexed field
# A key whose NgramType (low 3 bits of key_hash) is not SingleTerm: its postings are NEVER embedded (index_posting.rs:445: the
# embedding arm asks for NgramType::SingleTerm) and every record starts with the positions count of each COMPONENT term in this
# doc -- field_vec_ngram1, field_vec_ngram2 (, field_vec_ngram3), each through write_field_vec (index_posting.rs:666-722), which
# for one indexed field is the count VINT of a SingleTerm record (848-872) -- BEFORE the key's own field vector (724-732: its
# positions_count) and its positions (734-741: positions_compressed).  The key's positions are the places of its FIRST word
# (tokenizer.rs:699: "position as u16 - 1" for a bigram).  Readers: the component tfs decode_positions_multiterm_singlefield
# add_result.rs:2074-2086 (scored by the n-gram arms of get_bm25f_multiterm_singlefield 1454-1477), the count 2089, the positions
# get_next_position_singlefield in the phrase check 3596-3684.
# ---------------------------------------------------------------------------------------------------------------- H8
# NgramType::NgramFF (two components), CompressionType::Array, pivot inside the list: p0, p1 2-byte pointers, p2 a 3-byte pointer
#   p0 doc 4     component tfs (3, 130)   key positions {7, 20}     130 needs the 2-byte count form
#   p1 doc 90    component tfs (1, 1)     key positions {0}
#   p2 doc 7000  component tfs (5, 2)     key positions {300, 301}  300 needs the 2-byte position form
H8_REC_P0 = [
    3 | STOP,                          # field_vec_ngram1 = [(0, 3)]: write_field_vec 858-860
    130 >> 7, (130 & 0x7F) | STOP,     # field_vec_ngram2 = [(0, 130)]: 861-865
    2 | STOP,                          # the key's own positions_count 2 (724-732)
    7 | STOP, 12 | STOP,               # positions 7, then 20 - 7 - 1 = 12 (compress_positions 955-957)
]                                      # 6 bytes
H8_REC_P1 = [1 | STOP, 1 | STOP, 1 | STOP, 0 | STOP]                       # tfs 1, 1; count 1; position 0 -> 4 bytes
H8_REC_P2 = [5 | STOP, 2 | STOP, 2 | STOP, 300 >> 7, (300 & 0x7F) | STOP, 0 | STOP]  # tfs 5, 2; count 2; 300 (958-965), 301 - 300 - 1 = 0 -> 6 bytes
H8_PREFIX = [0x42, 0x43]
H8_R = len(H8_PREFIX) + len(H8_REC_P2) + len(H8_REC_P1) + len(H8_REC_P0)   # = 18: p0's record touches R, the later ones lie below
H8_POINTERS = [
    6, 0,                              # p0: running record size 6, 2 bytes, top bit clear (compress_postinglist.rs:477-482)
    10, 0,                             # p1: 6 + 4
    16, 0, 0,                          # p2: 10 + 6, 3 bytes (507-515)
]
H8 = dict(
    name="H8 bigram key, component tfs before the key's positions",
    n_components=2,
    block_id=1, compression_type_pointer=(1 << 30) | H8_R, posting_count=3, pointer_pivot_p_docid=2,
    body=bytes(H8_PREFIX + H8_REC_P2 + H8_REC_P1 + H8_REC_P0 + H8_POINTERS + _u16(4) + _u16(90) + _u16(7000)),
    docs=[4, 90, 7000], component_tfs=[[3, 130], [1, 1], [5, 2]], counts=[2, 1, 2], positions=[7, 20] + [0] + [300, 301])
# ---------------------------------------------------------------------------------------------------------------- H9
# NgramType::NgramFFF (three components), CompressionType::Rle, one run of two docs, 2-byte pointers
#   p0 doc 500   component tfs (2, 9, 200)   key positions {1}
#   p1 doc 501   component tfs (1, 1, 1)     key positions {40, 41, 42}
H9_REC_P0 = [2 | STOP, 9 | STOP, 200 >> 7, (200 & 0x7F) | STOP, 1 | STOP, 1 | STOP]   # 6 bytes
H9_REC_P1 = [1 | STOP, 1 | STOP, 1 | STOP, 3 | STOP, 40 | STOP, 0 | STOP, 0 | STOP]  # 7 bytes
H9_R = len(H9_REC_P1) + len(H9_REC_P0)                                                # = 13
H9 = dict(
    name="H9 trigram key, rle container",
    n_components=3,
    block_id=0, compression_type_pointer=(3 << 30) | H9_R, posting_count=2, pointer_pivot_p_docid=2,
    body=bytes(H9_REC_P1 + H9_REC_P0 + [6, 0, 13, 0] + _u16(1) + _u16(500) + _u16(1)),   # Rle: 1 run, start 500, one doc after it
    docs=[500, 501], component_tfs=[[2, 9, 200], [1, 1, 1]], counts=[1, 3], positions=[1] + [40, 41, 42])

BLOCKS = [H1, H2, H3, H4, H7]
FIELD_BLOCKS = [H5, H6]
NGRAM_BLOCKS = [H8, H9]
