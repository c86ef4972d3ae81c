// Reader of the reference's in-RAM posting-list block format (first step): the structures a
// SeekStorm shard holds after open_shard -- BlockObjectIndex (index.rs:781-789) + the segment's key-body byte array
// (index.rs:991-995) -- are decoded on the host into (doc id, tf) postings and handed to the normal image builder.
//
// Layout of one posting list inside the key-body slice (compress_postinglist.rs:694-946, intersection.rs:211-226):
//
//     [ ... position records (VINT) ... ] <- rank_position_pointer_range = compression_type_pointer & 0x3FFF_FFFF
//     [ rank/position pointers: 2 bytes each for rank < pointer_pivot_p_docid, 3 bytes each from there on ]
//     [ doc-id container: Array n x u16 | Bitmap 8192 B | Rle u16 runs + runs x (u16 start, u16 length) ]
//
// tf of a posting = positions_count of its pointer (add_result.rs:2036-2197, single indexed field, SingleTerm keys):
// embedded pointers carry the count in their top bits, the others point backwards into the VINT area whose first
// value is the count (read_singlefield_value, add_result.rs:2584-2606).  N-gram keys put the tf of each component term
// before the count (ss_ref_decode_block_ngram); several indexed fields: ss_ref_decode_block_fields below.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <memory>
#include <vector>

#include "ss_common.h"
#include "ss_threads.h"

namespace {

inline uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t rd24(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }

// read_singlefield_value (add_result.rs:2584-2606): 1-3 bytes, 7 bits each, most significant group first, the LAST
// byte carries the stop bit 0x80
inline bool read_vint(const uint8_t* a, uint64_t len, uint64_t pos, uint32_t* out) {
  if (pos >= len) return false;
  uint32_t v = a[pos];
  if (v & 0x80u) { *out = v & 0x7Fu; return true; }
  if (pos + 1 >= len) return false;
  const uint32_t b2 = a[pos + 1];
  if (b2 & 0x80u) { *out = ((v & 0x7Fu) << 7) | (b2 & 0x7Fu); return true; }
  if (pos + 2 >= len) return false;
  *out = ((v & 0x7Fu) << 14) | ((b2 & 0x7Fu) << 7) | (a[pos + 2] & 0x7Fu);
  return true;
}

// a POSITION of a record (compress_positions, compress_postinglist.rs:948-976; get_next_position_singlefield / _multifield,
// add_result.rs:36-88): one and two bytes like the VINT above, but the THREE-byte form keeps bit 13 twice -- the writer stores
// delta >> 13, (delta >> 7) & 0x7F, delta & 0x7F and the reader ORs b0 << 13 | b1 << 7 | b2 -- so it is not the count's VINT
inline bool read_position(const uint8_t* a, uint64_t len, uint64_t pos, uint32_t* out) {
  if (pos >= len) return false;
  const uint32_t v = a[pos];
  if (v & 0x80u) { *out = v & 0x7Fu; return true; }
  if (pos + 1 >= len) return false;
  const uint32_t b2 = a[pos + 1];
  if (b2 & 0x80u) { *out = (v << 7) | (b2 & 0x7Fu); return true; }
  if (pos + 2 >= len) return false;
  *out = (v << 13) | (b2 << 7) | (a[pos + 2] & 0x7Fu);
  return true;
}

}  // namespace

namespace {
// pos_out (optional): the positions of every posting appended in posting order, ABSOLUTE (the file stores the first position
// and then gap - 1: get_next_position_singlefield + 1, add_result.rs:3596-3684); a position beyond 65 535 -> SS_ENOTSUP
// N-gram keys (n_components 2 / 3): the record's positions are the KEY's own (the positions of its first word, tokenizer.rs:699:
// "position - 1"); they are handed out with component 0 and counted in npos_out, the other components get none.
// npos_out (with pos_out): positions per posting -- the tf for a SingleTerm key, the key's positions_count / 0 for n-gram components.
int decode_block(const ss_ref_block* b, uint32_t n_components, uint32_t component, uint16_t* docs_out, uint16_t* tfs_out,
                 std::vector<uint16_t>* pos_out = nullptr, std::vector<uint16_t>* npos_out = nullptr);
}
// Decodes one block.  docs_out / tfs_out need room for 65 536 entries.  Returns the posting count or a negative code.
extern "C" int ss_ref_decode_block(const ss_ref_block* b, uint16_t* docs_out, uint16_t* tfs_out) {
  return decode_block(b, 1, 0, docs_out, tfs_out);
}
// The same with the positions of every posting (SingleTerm keys, one indexed field): what a phrase query walks
// (decode_positions_multiterm_singlefield's embedded_positions / positions_pointer, add_result.rs:2036-2197, consumed by
// get_next_position_singlefield).  pos_out receives sum(tf) absolute positions in posting order; *n_pos_out = that sum
// (SS_EINVAL with the needed size when pos_cap is too small).
extern "C" int ss_ref_decode_block_positions(const ss_ref_block* b, uint16_t* docs_out, uint16_t* tfs_out, uint16_t* pos_out,
                                             uint64_t pos_cap, uint64_t* n_pos_out) {
  if (!n_pos_out) return SS_EINVAL;
  std::vector<uint16_t> pos;
  const int n = decode_block(b, 1, 0, docs_out, tfs_out, &pos);
  if (n < 0) return n;
  *n_pos_out = pos.size();
  if (pos.size() > pos_cap) return SS_EINVAL;
  if (pos_out && !pos.empty()) std::memcpy(pos_out, pos.data(), pos.size() * sizeof(uint16_t));
  return n;
}
// The same for a block of an N-GRAM key (NgramType != SingleTerm, index.rs:1854-1872; one indexed field): its postings are
// never embedded (index_posting.rs:445) and every record starts with the tf of each component term -- 2 for the bigram
// types, 3 for the trigram types -- before the positions count (decode_positions_multiterm_singlefield,
// add_result.rs:2074-2089).  tfs_out = tf of component `component` (0-based), which is what the n-gram arms of
// get_bm25f_multiterm_singlefield (add_result.rs:1454-1477) score with idf_ngram{1,2,3}.
extern "C" int ss_ref_decode_block_ngram(const ss_ref_block* b, uint32_t n_components, uint32_t component, uint16_t* docs_out,
                                         uint16_t* tfs_out) {
  if (n_components < 2 || n_components > 3 || component >= n_components) return SS_EINVAL;
  return decode_block(b, n_components, component, docs_out, tfs_out);
}
// ... and the n-gram key's OWN positions (the record's positions_count and positions behind the component tfs: what the phrase
// check walks for a query term that resolved to the key, add_result.rs:2074-2089, 3596-3684): npos_out [65536] = positions per
// posting, pos_out = their concatenation.  tfs_out = tf of component 0.
extern "C" int ss_ref_decode_block_ngram_positions(const ss_ref_block* b, uint32_t n_components, uint16_t* docs_out, uint16_t* tfs_out,
                                                   uint16_t* npos_out, uint16_t* pos_out, uint64_t pos_cap, uint64_t* n_pos_out) {
  if (n_components < 2 || n_components > 3 || !npos_out || !n_pos_out) return SS_EINVAL;
  std::vector<uint16_t> pos, np;
  const int n = decode_block(b, n_components, 0, docs_out, tfs_out, &pos, &np);
  if (n < 0) return n;
  *n_pos_out = pos.size();
  if (pos.size() > pos_cap) return SS_EINVAL;
  if (pos_out && !pos.empty()) std::memcpy(pos_out, pos.data(), pos.size() * sizeof(uint16_t));
  std::memcpy(npos_out, np.data(), np.size() * sizeof(uint16_t));
  return n;
}
namespace {
int decode_block(const ss_ref_block* b, uint32_t n_components, uint32_t component, uint16_t* docs_out, uint16_t* tfs_out,
                 std::vector<uint16_t>* pos_out, std::vector<uint16_t>* npos_out) {
  if (!b || !b->byte_array || !docs_out || !tfs_out) return SS_EINVAL;
  const bool ngram = n_components > 1;
  if (pos_out && ngram && !npos_out) return SS_EINVAL;  // an n-gram key's positions are the key's, not its components': counted apart
  const uint8_t* a = b->byte_array;
  const uint64_t len = b->byte_array_len;
  // a position takes at least one byte of its record, an embedded pointer holds at most 4: a block cannot carry more positions
  // than that -- corrupted pointers that send many postings to the same bytes are refused instead of decoded 65 536 times
  const size_t pos_base = pos_out ? pos_out->size() : 0;
  const uint64_t pos_cap = len + 4ull * 65536ull;
  const uint32_t ctype = b->compression_type_pointer >> 30;                 // CompressionType, index.rs:838-843
  const uint64_t range = b->compression_type_pointer & 0x3FFFFFFFu;         // rank_position_pointer_range
  const uint32_t count = (uint32_t)b->posting_count_m1 + 1u;
  const uint32_t pivot = b->pointer_pivot_p_docid;
  // pointer array size (intersection.rs:219-226)
  const uint64_t ptr_bytes = (uint64_t)pivot * 2u + (pivot <= b->posting_count_m1 ? (uint64_t)(count - pivot) * 3u : 0u);
  const uint64_t cont = range + ptr_bytes;  // compressed_doc_id_range
  if (cont > len) return SS_EINVAL;

  // ---- doc-id container
  uint32_t n = 0;
  if (ctype == 1u) {  // Array: count x u16 LE, ascending
    if (cont + (uint64_t)count * 2u > len) return SS_EINVAL;
    for (uint32_t i = 0; i < count; i++) docs_out[n++] = (uint16_t)rd16(a + cont + 2u * i);
  } else if (ctype == 2u) {  // Bitmap: bit d <-> byte d >> 3, bit d & 7 (compress_postinglist.rs:818-823)
    if (cont + 8192u > len) return SS_EINVAL;
    for (uint32_t w = 0; w < 1024u && n <= 65536u; w++) {
      uint64_t x;
      std::memcpy(&x, a + cont + 8u * w, 8);
      while (x) {
        if (n >= 65536u) return SS_EINVAL;
        docs_out[n++] = (uint16_t)(w * 64u + (uint32_t)__builtin_ctzll(x));
        x &= x - 1;
      }
    }
  } else if (ctype == 3u) {  // Rle: u16 runs, then (u16 start, u16 run_length): docs start ..= start + run_length
    if (cont + 2u > len) return SS_EINVAL;
    const uint32_t runs = rd16(a + cont);
    if (cont + 2u + (uint64_t)runs * 4u > len) return SS_EINVAL;
    for (uint32_t r = 0; r < runs; r++) {
      const uint32_t s = rd16(a + cont + 2u + 4u * r), l = rd16(a + cont + 4u + 4u * r);
      for (uint32_t j = 0; j <= l; j++) {
        if (n >= 65536u || s + j > 65535u) return SS_EINVAL;
        docs_out[n++] = (uint16_t)(s + j);
      }
    }
  } else {
    return SS_ENOTSUP;  // Delta: its writer is disabled in the reference (compress_postinglist.rs:242)
  }
  if (n != count) return SS_EINVAL;
  for (uint32_t i = 1; i < n; i++)
    if (docs_out[i] <= docs_out[i - 1]) return SS_EINVAL;

  // ---- tf from the rank/position pointers (add_result.rs:2044-2180)
  // a record behind a non-embedded pointer: [n-gram keys: tf of each component,] positions count, positions
  auto record_tf = [&](uint64_t back, uint32_t* tf) -> bool {
    if (back > range) return false;
    uint64_t pos = range - back;
    for (uint32_t c = 0; c < (ngram ? n_components : 1u); c++) {
      uint32_t v;
      if (!read_vint(a, len, pos, &v)) return false;
      if (!ngram || c == component) { *tf = v; return true; }
      pos += a[pos] & 0x80u ? 1u : (a[pos + 1] & 0x80u ? 2u : 3u);
    }
    return false;
  };
  // the positions of a record behind a non-embedded pointer: [n-gram keys: the component tfs,] the count, then that many VINTs,
  // each "gap - 1" after the first
  auto record_positions = [&](uint64_t back, uint32_t* count_out) -> int {
    uint64_t pos = range - back;
    uint32_t v, at = 0, tf = 0;
    for (uint32_t c = 0; c <= (ngram ? n_components : 0u); c++) {  // (component tfs,) the count
      if (!read_vint(a, len, pos, &v)) return SS_EINVAL;
      pos += a[pos] & 0x80u ? 1u : (a[pos + 1] & 0x80u ? 2u : 3u);
      tf = v;
    }
    if (tf == 0u || tf > 65535u) return SS_EINVAL;
    *count_out = tf;
    for (uint32_t i = 0; i < tf; i++) {
      if (!read_position(a, len, pos, &v)) return SS_EINVAL;
      pos += a[pos] & 0x80u ? 1u : (a[pos + 1] & 0x80u ? 2u : 3u);
      at = i == 0 ? v : at + v + 1u;
      if (at > 65535u) return SS_ENOTSUP;
      if (pos_out->size() - pos_base >= pos_cap) return SS_EINVAL;  // more positions than the bytes can hold: overlapping records
      pos_out->push_back((uint16_t)at);
    }
    return SS_OK;
  };
  // the positions inside an embedded pointer: widths as decode_positions_multiterm_singlefield unpacks them
  auto embedded_positions = [&](const uint32_t* d, uint32_t tf) -> int {
    uint32_t at = 0;
    for (uint32_t i = 0; i < tf; i++) {
      at = i == 0 ? d[0] : at + d[i] + 1u;
      if (at > 65535u) return SS_ENOTSUP;
      pos_out->push_back((uint16_t)at);
    }
    return SS_OK;
  };
  for (uint32_t r = 0; r < count; r++) {
    uint32_t tf = 0, np_r = 0xFFFFFFFFu;  // np_r: positions of a recorded posting (embedded ones carry their tf)
    if (r < pivot) {
      const uint64_t at = range + (uint64_t)r * 2u;
      if (at + 2u > len) return SS_EINVAL;
      const uint32_t p = rd16(a + at);
      if (p & 0x8000u) {  // embedded: 10 -> one position, 11 -> two
        if (ngram) return SS_EINVAL;
        const uint32_t tag = p >> 14;
        tf = tag == 2u ? 1u : tag == 3u ? 2u : 0u;
        if (pos_out && tf) {
          const uint32_t d1[2] = {p & 0x3FFFu, 0u}, d2[2] = {(p >> 7) & 0x7Fu, p & 0x7Fu};
          const int rc = embedded_positions(tf == 1u ? d1 : d2, tf);
          if (rc) return rc;
        }
      } else if (!record_tf(p & 0x7FFFu, &tf)) {
        return SS_EINVAL;
      } else if (pos_out && (!ngram || component == 0u)) {
        const int rc = record_positions(p & 0x7FFFu, &np_r);
        if (rc) return rc;
      }
    } else {
      const uint64_t at = range + (uint64_t)r * 3u - pivot;
      if (at + 3u > len) return SS_EINVAL;
      const uint32_t p = rd24(a + at);
      if (p & 0x800000u) {  // embedded: 100 / 101 / 110 / 111 -> 1..4 positions
        if (ngram) return SS_EINVAL;
        const uint32_t tag = p >> 21;
        tf = tag >= 4u ? tag - 3u : 0u;
        if (pos_out && tf) {
          const uint32_t d1[4] = {p & 0x1FFFFFu, 0u, 0u, 0u}, d2[4] = {(p >> 11) & 0x3FFu, p & 0x7FFu, 0u, 0u};
          const uint32_t d3[4] = {(p >> 14) & 0x7Fu, (p >> 7) & 0x7Fu, p & 0x7Fu, 0u};
          const uint32_t d4[4] = {(p >> 16) & 0x1Fu, (p >> 11) & 0x1Fu, (p >> 6) & 0x1Fu, p & 0x3Fu};
          const int rc = embedded_positions(tf == 1u ? d1 : tf == 2u ? d2 : tf == 3u ? d3 : d4, tf);
          if (rc) return rc;
        }
      } else if (!record_tf(p & 0x7FFFFFu, &tf)) {
        return SS_EINVAL;
      } else if (pos_out && (!ngram || component == 0u)) {
        const int rc = record_positions(p & 0x7FFFFFu, &np_r);
        if (rc) return rc;
      }
    }
    if (tf == 0u || tf > 65535u) return SS_EINVAL;  // positions_count >= 1 always; component tfs likewise
    tfs_out[r] = (uint16_t)tf;
    if (pos_out && npos_out) npos_out->push_back((uint16_t)(ngram ? (component == 0u ? np_r : 0u) : tf));
  }
  return (int)count;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// Several indexed fields: a posting's field vector [(field id, positions_count)], SingleTerm keys
// (decode_positions_multiterm_multifield add_result.rs:1485-2034, read_multifield_vec 2200-2293; writer
// index_posting.rs:445-660, write_field_vec 846-940).  indexed_field_id_bits = bit length of (field count - 1)
// (index.rs:2569-2570); longest_field_id from the first level of index.bin (commit.rs:264-274).
namespace {
struct FieldEntry { uint8_t field; uint32_t tf; };

// read_multifield_vec, more than one indexed field (add_result.rs:2229-2293)
bool read_field_vec(const uint8_t* a, uint64_t len, uint64_t pos, uint32_t id_bits, uint32_t longest, FieldEntry* out, int* n_out,
                    uint64_t* end_out = nullptr) {
  if (pos >= len) return false;
  int n = 0;
  if (a[pos] & 0x40u) {  // only the longest field: count in 6 + 7 (+ 7) bits
    uint32_t c = a[pos++];
    if (c & 0x80u) c &= 0x3Fu;
    else {
      if (pos >= len) return false;
      c = (c & 0x3Fu) << 7;
      const uint32_t c2 = a[pos++];
      if (c2 & 0x80u) c |= c2 & 0x7Fu;
      else {
        if (pos >= len) return false;
        c = (c << 7) | ((c2 & 0x7Fu) << 7) | (a[pos++] & 0x7Fu);
      }
    }
    out[n].field = (uint8_t)longest; out[n].tf = c; n++;
  } else {
    bool first = true;
    for (;;) {
      if (pos >= len || n >= 8) return false;
      uint32_t b = a[pos++];
      const bool field_stop = (b & (first ? 0x20u : 0x40u)) != 0;  // FIELD_STOP_BIT_1 / _2 (index.rs:112-113)
      uint32_t v = b & (first ? 0x1Fu : 0x3Fu);
      if (!(b & 0x80u)) {
        if (pos >= len) return false;
        b = a[pos++];
        v = (v << 7) | (b & 0x7Fu);
        if (!(b & 0x80u)) {
          if (pos >= len) return false;
          b = a[pos++];
          v = (v << 7) | (b & 0x7Fu);
        }
      }
      out[n].field = (uint8_t)(v & ((1u << id_bits) - 1u)); out[n].tf = v >> id_bits; n++;
      first = false;
      if ((b & 0x80u) && field_stop) break;
    }
  }
  *n_out = n;
  if (end_out) *end_out = pos;
  return true;
}
// npos_out (with pos_out): positions per ENTRY -- its tf for a SingleTerm key; for an n-gram key the key's OWN count in the entry's
// field with component 0 (the key's field vector and positions follow the components' vectors in the record), 0 with the others
int decode_block_fields(const ss_ref_block* b, uint32_t n_fields, uint32_t longest_field_id, uint32_t n_components, uint32_t component,
                        uint16_t* docs_out, uint32_t* first_out, uint8_t* field_out, uint16_t* tf_out,
                        std::vector<uint16_t>* pos_out = nullptr, std::vector<uint16_t>* npos_out = nullptr);
}  // namespace

// Decodes one block of a multi-field index: docs_out [65536], first_out [65537] = CSR of the field entries per posting,
// field_out / tf_out [65536 * n_fields].  Returns the posting count or a negative code.
extern "C" int ss_ref_decode_block_fields(const ss_ref_block* b, uint32_t n_fields, uint32_t longest_field_id, uint16_t* docs_out,
                                          uint32_t* first_out, uint8_t* field_out, uint16_t* tf_out) {
  return decode_block_fields(b, n_fields, longest_field_id, 1, 0, docs_out, first_out, field_out, tf_out);
}
// ... with the positions of every (posting, field) entry in entry order (sum of tf_out values; SingleTerm keys): what the phrase
// check of add_result_multiterm_multifield reads through get_next_position_multifield.  Returns the posting count.
extern "C" int ss_ref_decode_block_fields_positions(const ss_ref_block* b, uint32_t n_fields, uint32_t longest_field_id, uint16_t* docs_out,
                                                    uint32_t* first_out, uint8_t* field_out, uint16_t* tf_out, uint16_t* pos_out,
                                                    uint64_t pos_cap, uint64_t* n_pos_out) {
  if (!n_pos_out || (pos_cap && !pos_out)) return SS_EINVAL;
  std::vector<uint16_t> pos;
  const int n = decode_block_fields(b, n_fields, longest_field_id, 1, 0, docs_out, first_out, field_out, tf_out, &pos);
  if (n < 0) return n;
  *n_pos_out = pos.size();
  if (pos.size() > pos_cap) return SS_EINVAL;
  if (!pos.empty()) std::memcpy(pos_out, pos.data(), pos.size() * sizeof(uint16_t));
  return n;
}
// The same for a block of an N-GRAM key of a multi-field index: never embedded, and every record starts with the field
// vector of each component term -- 2 for the bigram types, 3 for the trigram types -- before the n-gram's own
// (index_posting.rs:664-722 writes them with write_field_vec one after the other; decode_positions_multiterm_multifield
// reads them into field_vec_ngram1..3, add_result.rs:1524-1600).  Output = the field vector of component `component`: what
// the n-gram arms of get_bm25f_multiterm_multifield (add_result.rs:1171-1426) score with idf_ngram{1,2,3}.
extern "C" int ss_ref_decode_block_fields_ngram(const ss_ref_block* b, uint32_t n_fields, uint32_t longest_field_id,
                                                uint32_t n_components, uint32_t component, uint16_t* docs_out, uint32_t* first_out,
                                                uint8_t* field_out, uint16_t* tf_out) {
  if (n_components < 2 || n_components > 3 || component >= n_components) return SS_EINVAL;
  return decode_block_fields(b, n_fields, longest_field_id, n_components, component, docs_out, first_out, field_out, tf_out);
}
// ... and the key's OWN positions of a multi-field n-gram block: the entries are those of component 0 (field, the component's tf);
// npos_out [65536 * n_fields] = the key's positions behind every entry (0 where the key does not stand in that field), pos_out their
// concatenation, field by field inside a posting.
extern "C" int ss_ref_decode_block_fields_ngram_positions(const ss_ref_block* b, uint32_t n_fields, uint32_t longest_field_id,
                                                          uint32_t n_components, uint16_t* docs_out, uint32_t* first_out, uint8_t* field_out,
                                                          uint16_t* tf_out, uint16_t* npos_out, uint16_t* pos_out, uint64_t pos_cap,
                                                          uint64_t* n_pos_out) {
  if (n_components < 2 || n_components > 3 || !npos_out || !n_pos_out || (pos_cap && !pos_out)) return SS_EINVAL;
  std::vector<uint16_t> pos, np;
  const int n = decode_block_fields(b, n_fields, longest_field_id, n_components, 0, docs_out, first_out, field_out, tf_out, &pos, &np);
  if (n < 0) return n;
  *n_pos_out = pos.size();
  if (pos.size() > pos_cap) return SS_EINVAL;
  if (!pos.empty()) std::memcpy(pos_out, pos.data(), pos.size() * sizeof(uint16_t));
  if (!np.empty()) std::memcpy(npos_out, np.data(), np.size() * sizeof(uint16_t));
  return n;
}
namespace {
int decode_block_fields(const ss_ref_block* b, uint32_t n_fields, uint32_t longest_field_id, uint32_t n_components, uint32_t component,
                        uint16_t* docs_out, uint32_t* first_out, uint8_t* field_out, uint16_t* tf_out, std::vector<uint16_t>* pos_out,
                        std::vector<uint16_t>* npos_out) {
  if (pos_out && n_components > 1 && !npos_out) return SS_EINVAL;  // an n-gram key's positions are the key's own: counted apart
  const size_t pos_base = pos_out ? pos_out->size() : 0;
  if (!b || !b->byte_array || !docs_out || !first_out || !field_out || !tf_out || n_fields < 2 || n_fields > 8 ||
      longest_field_id >= n_fields) return SS_EINVAL;
  const bool ngram = n_components > 1;
  // doc ids: the container walk of the single-field reader (tf output unused); run on a one-field view of the pointers
  // would misread them, so walk the container here again
  const uint8_t* a = b->byte_array;
  const uint64_t len = b->byte_array_len;
  const uint32_t ctype = b->compression_type_pointer >> 30;
  const uint64_t range = b->compression_type_pointer & 0x3FFFFFFFu;
  const uint32_t count = (uint32_t)b->posting_count_m1 + 1u;
  const uint32_t pivot = b->pointer_pivot_p_docid;
  const uint64_t ptr_bytes = (uint64_t)pivot * 2u + (pivot <= b->posting_count_m1 ? (uint64_t)(count - pivot) * 3u : 0u);
  const uint64_t cont = range + ptr_bytes;
  if (cont > len) return SS_EINVAL;
  uint32_t n = 0;
  if (ctype == 1u) {
    if (cont + (uint64_t)count * 2u > len) return SS_EINVAL;
    for (uint32_t i = 0; i < count; i++) docs_out[n++] = (uint16_t)rd16(a + cont + 2u * i);
  } else if (ctype == 2u) {
    if (cont + 8192u > len) return SS_EINVAL;
    for (uint32_t w = 0; w < 1024u; w++) {
      uint64_t x;
      std::memcpy(&x, a + cont + 8u * w, 8);
      while (x) {
        if (n >= 65536u) return SS_EINVAL;
        docs_out[n++] = (uint16_t)(w * 64u + (uint32_t)__builtin_ctzll(x));
        x &= x - 1;
      }
    }
  } else if (ctype == 3u) {
    if (cont + 2u > len) return SS_EINVAL;
    const uint32_t runs = rd16(a + cont);
    if (cont + 2u + (uint64_t)runs * 4u > len) return SS_EINVAL;
    for (uint32_t r = 0; r < runs; r++) {
      const uint32_t st = rd16(a + cont + 2u + 4u * r), l = rd16(a + cont + 4u + 4u * r);
      for (uint32_t j = 0; j <= l; j++) {
        if (n >= 65536u || st + j > 65535u) return SS_EINVAL;
        docs_out[n++] = (uint16_t)(st + j);
      }
    }
  } else {
    return SS_ENOTSUP;
  }
  if (n != count) return SS_EINVAL;
  for (uint32_t i = 1; i < n; i++)
    if (docs_out[i] <= docs_out[i - 1]) return SS_EINVAL;

  uint32_t id_bits = 0;
  while ((1u << id_bits) < n_fields) id_bits++;  // usize::BITS - (len - 1).leading_zeros()
  const uint32_t id_mask = (1u << id_bits) - 1u;
  uint32_t w = 0;
  for (uint32_t r = 0; r < count; r++) {
    first_out[r] = w;
    FieldEntry e[8];
    int ne = 0;
    uint32_t own_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // positions handed out per entry (recorded postings; embedded ones: their tf)
    bool recorded = false;
    uint32_t emb_bits = 0;  // embedded pointer: bits that hold the positions
    const bool two = r < pivot;
    const uint64_t at = two ? range + (uint64_t)r * 2u : range + (uint64_t)r * 3u - pivot;
    if (at + (two ? 2u : 3u) > len) return SS_EINVAL;
    const uint32_t p = two ? rd16(a + at) : rd24(a + at);
    if (!(p & (two ? 0x8000u : 0x800000u))) {  // record in the position area
      const uint64_t back = p & (two ? 0x7FFFu : 0x7FFFFFu);
      if (back > range) return SS_EINVAL;
      recorded = true;
      uint64_t at_rec = range - back;
      for (uint32_t c = 0; c <= (ngram ? component : 0u); c++) {  // n-gram keys: the components' vectors come first
        if (!read_field_vec(a, len, at_rec, id_bits, longest_field_id, e, &ne, &at_rec)) return SS_EINVAL;
      }
      // whose positions follow: the posting's own vector (SingleTerm) / the key's own vector behind ALL component vectors (n-gram
      // key, handed out with component 0: index_posting.rs:666-741)
      FieldEntry own[8];
      int nown = 0;
      if (pos_out && ngram && component == 0u) {
        FieldEntry skip[8];
        int nskip = 0;
        for (uint32_t c = 1; c < n_components; c++)
          if (!read_field_vec(a, len, at_rec, id_bits, longest_field_id, skip, &nskip, &at_rec)) return SS_EINVAL;
        if (!read_field_vec(a, len, at_rec, id_bits, longest_field_id, own, &nown, &at_rec)) return SS_EINVAL;
        for (int i = 0, k = 0; i < nown; i++) {  // the key stands in a field only where its first word does
          while (k < ne && e[k].field < own[i].field) k++;
          if (k == ne || e[k].field != own[i].field || own[i].tf == 0 || own[i].tf > 65535u) return SS_EINVAL;
        }
      } else if (pos_out && !ngram) {
        nown = ne;
        for (int i = 0; i < ne; i++) own[i] = e[i];
      }
      for (int i = 0; i < ne; i++) own_cnt[i] = 0;
      for (int i = 0; i < nown; i++)
        for (int k = 0; k < ne; k++)
          if (e[k].field == own[i].field) own_cnt[k] = own[i].tf;
      if (pos_out) {  // the positions follow the field vector: per field its tf VINTs, the first absolute, then "gap - 1"
        for (int i = 0; i < nown; i++) {  // (get_next_position_multifield restarts at every field, add_result.rs:3279-3283)
          uint32_t at_pos = 0, v;
          for (uint32_t x = 0; x < own[i].tf; x++) {
            if (!read_position(a, len, at_rec, &v)) return SS_EINVAL;
            at_rec += a[at_rec] & 0x80u ? 1u : (a[at_rec + 1] & 0x80u ? 2u : 3u);
            at_pos = x == 0 ? v : at_pos + v + 1u;
            if (at_pos > 65535u) return SS_ENOTSUP;
            if (pos_out->size() - pos_base >= len + 4ull * 65536ull) return SS_EINVAL;  // overlapping records (corrupted pointers)
            pos_out->push_back((uint16_t)at_pos);
          }
        }
      }
    } else if (ngram) {
      return SS_EINVAL;  // n-gram postings are never embedded (index_posting.rs:445)
    } else if (two) {  // embedded, 2 bytes: tag = bits 15..12 (add_result.rs:1606-1737)
      const uint32_t tag = p >> 12, pb = 12u - id_bits;
      switch (tag) {
        case 0xC: case 0xD: e[0] = {(uint8_t)longest_field_id, 1}; ne = 1; emb_bits = 13u; break;
        case 0xE: case 0xF: e[0] = {(uint8_t)longest_field_id, 2}; ne = 1; emb_bits = 13u; break;
        case 0x8: case 0x9: case 0xA: e[0] = {(uint8_t)((p >> pb) & id_mask), tag - 7u}; ne = 1; emb_bits = pb; break;
        case 0xB: {
          const uint32_t pb2 = 12u - 2u * id_bits;
          e[0] = {(uint8_t)((p >> (pb2 + id_bits)) & id_mask), 1};
          e[1] = {(uint8_t)((p >> pb2) & id_mask), 1};
          ne = 2;
          emb_bits = pb2;
          break;
        }
        default: return SS_EINVAL;
      }
    } else {  // embedded, 3 bytes: tag = bits 23..19 (add_result.rs:1738-2017)
      const uint32_t tag = p >> 19, pb = 19u - id_bits, pb2 = 19u - 2u * id_bits, pb3 = 19u - 3u * id_bits;
      if (tag >= 0x18u) { e[0] = {(uint8_t)longest_field_id, ((tag - 0x18u) >> 1) + 1u}; ne = 1; emb_bits = 20u; }
      else if (tag >= 0x10u && tag <= 0x13u) { e[0] = {(uint8_t)((p >> pb) & id_mask), tag - 0x0Fu}; ne = 1; emb_bits = pb; }
      else if (tag >= 0x14u && tag <= 0x16u) {
        e[0] = {(uint8_t)((p >> (pb2 + id_bits)) & id_mask), tag == 0x16u ? 2u : 1u};
        e[1] = {(uint8_t)((p >> pb2) & id_mask), tag == 0x15u ? 2u : 1u};
        ne = 2;
        emb_bits = pb2;
      } else if (tag == 0x17u) {
        e[0] = {(uint8_t)((p >> (pb3 + 2u * id_bits)) & id_mask), 1};
        e[1] = {(uint8_t)((p >> (pb3 + id_bits)) & id_mask), 1};
        e[2] = {(uint8_t)((p >> pb3) & id_mask), 1};
        ne = 3;
        emb_bits = pb3;
      } else return SS_EINVAL;
    }
    if (pos_out && emb_bits) {
      // the embedded positions: the low emb_bits bits of the pointer hold all positions of the posting, field after field, as
      // "first absolute, then gap - 1" per field; the bits are dealt out front to back, position i of n getting
      // floor(bits left / (n - i)) -- every arm of add_result.rs:1617-2012 is an instance of that rule (13 -> 6 + 7,
      // 20 -> 10 + 10 / 6 + 7 + 7 / 5 + 5 + 5 + 5, position_bits -> /2, /3, /4 ...; writer index_posting.rs:592-660)
      uint32_t n_pos = 0, left = emb_bits, taken = 0;
      for (int i = 0; i < ne; i++) n_pos += e[i].tf;
      for (int i = 0; i < ne; i++) {
        uint32_t at_pos = 0;
        for (uint32_t x = 0; x < e[i].tf; x++, taken++) {
          const uint32_t wbits = left / (n_pos - taken);
          left -= wbits;
          const uint32_t v = (p >> left) & ((1u << wbits) - 1u);
          at_pos = x == 0 ? v : at_pos + v + 1u;
          if (at_pos > 65535u) return SS_ENOTSUP;
          pos_out->push_back((uint16_t)at_pos);
        }
      }
    }
    for (int i = 0; i < ne; i++) {
      if (e[i].field >= n_fields || e[i].tf == 0 || e[i].tf > 65535u) return SS_EINVAL;
      if (i && e[i].field <= e[i - 1].field) return SS_EINVAL;  // the field vector is written in ascending field order
      field_out[w] = e[i].field;
      tf_out[w] = (uint16_t)e[i].tf;
      if (pos_out && npos_out) npos_out->push_back((uint16_t)(recorded ? own_cnt[i] : e[i].tf));
      w++;
    }
  }
  first_out[count] = w;
  return (int)count;
}
}  // namespace

extern "C" int ss_bm25_upload_ref_blocks(ss_shard* s, uint64_t n_docs, const uint8_t* doclen_bytes, uint32_t n_terms,
                                         const uint64_t* term_block_offsets, const ss_ref_block* blocks) {
  if (!s || !doclen_bytes || !term_block_offsets || n_docs == 0 || n_terms == 0) return SS_EINVAL;
  if (term_block_offsets[n_terms] && !blocks) return SS_EINVAL;
  std::vector<uint64_t> offs((size_t)n_terms + 1, 0);
  std::vector<uint32_t> docs;
  std::vector<uint16_t> tfs;
  std::vector<uint16_t> d16(65536), t16(65536);
  for (uint32_t t = 0; t < n_terms; t++) {
    offs[t] = docs.size();
    uint64_t prev_block = 0;
    for (uint64_t bi = term_block_offsets[t]; bi < term_block_offsets[t + 1]; bi++) {
      const ss_ref_block& b = blocks[bi];
      if (bi > term_block_offsets[t] && b.block_id <= prev_block) return SS_EINVAL;  // blocks ascending by block_id
      prev_block = b.block_id;
      const int n = ss_ref_decode_block(&b, d16.data(), t16.data());
      if (n < 0) return n;
      for (int i = 0; i < n; i++) {
        const uint64_t doc = ((uint64_t)b.block_id << 16) | d16[i];  // index.rs:115 ROARING_BLOCK_SIZE = 65 536
        if (doc >= n_docs) return SS_EINVAL;
        docs.push_back((uint32_t)doc);
        tfs.push_back(t16[i]);
      }
    }
  }
  offs[n_terms] = docs.size();
  return ss_bm25_upload(s, n_docs, doclen_bytes, n_terms, offs.data(), docs.data(), tfs.data());
}

// ---------------------------------------------------------------------------------------------------------------
// index.bin walker (writer commit.rs:264-369 + commit_segment 467-552, reader index.rs:3263-3740)
//
//   u16 major (= 6), u16 minor                                                         index.rs:103-107, 2839-2853
//   per level (one 65 536-doc block id, shared by all keys):
//     [u16 longest_field_id]   first level only                                        commit.rs:264-274
//     indexed_field_count x [u8; 65536] length bytes                                   commit.rs:276-281
//     u64 indexed_doc_count (cumulative), u64 positions_sum_normalized (cumulative)    commit.rs:298-311
//     segment_number1 x (u32 block_length, u32 key_count)                              commit.rs:313-316, 355-363
//     per segment: key_count heads of key_head_size bytes, ascending key_hash, then the key bodies
//   key head (compress_postinglist.rs:339-409): u64 key_hash @0 (low 3 bits = NgramType), u16 posting_count - 1 @8,
//     u16 max_docid @10, u16 max_p_docid @12, n-gram df bytes @14.., u16 pointer_pivot_p_docid @size-6,
//     u32 compression_type_pointer @size-4
struct ss_index_bin {
  const uint8_t* bytes = nullptr;
  uint64_t len = 0;
  uint32_t n_fields = 1, key_head_size = 20, seg_bits = 11;
  uint64_t n_docs = 0, positions_sum = 0;
  uint32_t n_ngram_keys = 0;
  uint16_t longest_field_id = 0;
  std::vector<const uint8_t*> doclen;  // per level: indexed_field_count arrays of 65 536 bytes
  struct Blk {
    uint64_t key;
    ss_ref_block b;
    uint8_t n_comp, comp;              // n-gram key: one entry per component term (2 or 3), SingleTerm: 1 / 0
    uint8_t df_byte;                   // posting_count_ngram_{comp+1}_compressed of the key head
  };                                   // (trivially constructible: see NoInitAlloc)
  std::vector<Blk, NoInitAlloc<Blk>> blocks;  // sorted by (key, component, block_id)
  std::vector<uint64_t> keys;          // ascending; term id = index.  An n-gram key appears once per component, in order
  std::vector<uint8_t> term_comp, term_ncomp, term_df_byte;  // per term; df byte of the LAST level (commit.rs:646-653)
  std::vector<uint64_t> term_block_off;
  uint32_t n_dense = 0xFFFFFFFFu;      // ss_index_bin_tier: terms [0, n_dense) go to the dense image, the rest to the sparse tier
};

namespace {
inline uint64_t rd64(const uint8_t* p) { uint64_t v; std::memcpy(&v, p, 8); return v; }
inline uint32_t rd32(const uint8_t* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
}  // namespace

static int index_bin_open_impl(const uint8_t* bytes, uint64_t len, uint32_t indexed_field_count, uint32_t key_head_size,
                               uint32_t segment_number_bits, ss_index_bin** out);
extern "C" int ss_index_bin_open(const uint8_t* bytes, uint64_t len, uint32_t indexed_field_count, uint32_t key_head_size,
                                 uint32_t segment_number_bits, ss_index_bin** out) {
  // (the walk allocates per-key tables on worker threads: a std::bad_alloc of any of them comes back as SS_ENOMEM, ss_threads.h)
  return ss_guard([&] { return index_bin_open_impl(bytes, len, indexed_field_count, key_head_size, segment_number_bits, out); }, SS_ENOMEM, SS_EDEVICE);
}
static int index_bin_open_impl(const uint8_t* bytes, uint64_t len, uint32_t indexed_field_count, uint32_t key_head_size,
                               uint32_t segment_number_bits, ss_index_bin** out) {
  if (!bytes || !out || len < 4 || indexed_field_count == 0 || segment_number_bits > 16) return SS_EINVAL;
  if (key_head_size != 20 && key_head_size != 22 && key_head_size != 23) return SS_EINVAL;  // index.rs:2806-2812
  if (rd16(bytes) != 6u) return SS_ENOTSUP;  // INDEX_FORMAT_VERSION_MAJOR
  std::unique_ptr<ss_index_bin> ix(new ss_index_bin);
  static const bool trace = getenv("SS_LOAD_TRACE") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[load] open: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
    t_last = t;
  };
  ix->bytes = bytes; ix->len = len;
  ix->n_fields = indexed_field_count; ix->key_head_size = key_head_size; ix->seg_bits = segment_number_bits;
  const uint64_t nseg = 1ull << segment_number_bits;
  // ---- the level headers, one after the other (a level's size is only known from its segment head table): where every segment lies
  struct Seg { uint32_t level; uint64_t pos, block_length, key_count; };
  std::vector<Seg> segs;
  {
    uint64_t pos = 4;
    uint32_t level = 0;
    while (pos < len) {
      if (level == 0) {
        if (pos + 2 > len) return SS_EINVAL;
        ix->longest_field_id = (uint16_t)rd16(bytes + pos);
        pos += 2;
      }
      const uint64_t fixed = (uint64_t)indexed_field_count * 65536u + 16u + nseg * 8u;
      if (pos + fixed > len) return SS_EINVAL;
      ix->doclen.push_back(bytes + pos);
      pos += (uint64_t)indexed_field_count * 65536u;
      const uint64_t docs = rd64(bytes + pos), psum = rd64(bytes + pos + 8);
      if (docs < ix->n_docs || docs > ((uint64_t)level + 1) * 65536u || docs <= (uint64_t)level * 65536u) return SS_EINVAL;
      ix->n_docs = docs; ix->positions_sum = psum;
      pos += 16;
      const uint8_t* heads = bytes + pos;
      pos += nseg * 8u;
      for (uint64_t k0 = 0; k0 < nseg; k0++) {
        const uint64_t block_length = rd32(heads + 8 * k0), key_count = rd32(heads + 8 * k0 + 4);
        if (key_count * key_head_size > block_length || pos + block_length > len) return SS_EINVAL;
        if (key_count) segs.push_back(Seg{level, pos, block_length, key_count});
        pos += block_length;
      }
      level++;
    }
  }
  lap("level headers");
  // ---- the key heads of every segment, segments in parallel: one entry per (key, component, level), dealt into 256 buckets by the
  // key hash's top byte (hashes: even buckets), each bucket then sorted on its own -- the concatenation is the sorted whole
  constexpr unsigned NB_LOG2 = 8, NB = 1u << NB_LOG2;  // (more buckets: smaller sorts but a slower scatter -- 1024 / 4096 measured, no gain)
  constexpr size_t GRAIN = 16;  // segments per chunk of work
  const size_t n_chunks = (segs.size() + GRAIN - 1) / GRAIN;
  // pass 1: how many entries every chunk of segments sends to every bucket (the heads' keys alone are read); then every chunk knows
  // where its entries go in the block table, and pass 2 writes them there directly -- nothing is staged, nothing grows
  std::vector<uint32_t> cnt(n_chunks * NB, 0u), skipped(n_chunks, 0u);
  std::atomic<int> bad{0};
  auto n_components = [&](uint64_t key, bool* skip) {
    // NgramType (index.rs:1854-1872): 0 SingleTerm, 1-3 bigrams, 4-7 trigrams.  An n-gram key is kept as one posting
    // list per component term (same docs, the component's tf): scored with idf_ngram_i each, their sum is the
    // n-gram arm of get_bm25f_multiterm_singlefield (add_result.rs:1454-1477); several fields: the components' field vectors.
    const uint32_t ntype = (uint32_t)(key & 7u), n_comp = ntype == 0 ? 1u : ntype <= 3u ? 2u : 3u;
    *skip = ntype && n_comp > key_head_size - 20u;  // a head without room for the component df bytes
    return n_comp;
  };
  ss_parallel_for(segs.size(), GRAIN, [&](size_t a, size_t b, unsigned) {
    uint32_t* c = cnt.data() + (a / GRAIN) * NB;
    for (size_t si = a; si < b; si++) {
      const Seg& sg = segs[si];
      uint64_t prev = 0;
      for (uint64_t i = 0; i < sg.key_count; i++) {
        const uint64_t key = rd64(bytes + sg.pos + i * key_head_size);
        if (i && key <= prev) { bad.store(1); return; }  // heads are binary-searched by key_hash (search.rs:2310-2357)
        prev = key;
        bool skip;
        const uint32_t n_comp = n_components(key, &skip);
        if (skip) { skipped[a / GRAIN]++; continue; }
        c[key >> (64 - NB_LOG2)] += n_comp;
      }
    }
  });
  if (bad.load()) return SS_EINVAL;
  lap("key heads counted");
  for (size_t ch = 0; ch < n_chunks; ch++) ix->n_ngram_keys += skipped[ch];
  std::vector<uint64_t> boff(NB + 1, 0);
  std::vector<uint64_t> start(n_chunks * NB);
  for (unsigned bkt = 0; bkt < NB; bkt++) {
    uint64_t at = boff[bkt];
    for (size_t ch = 0; ch < n_chunks; ch++) { start[ch * NB + bkt] = at; at += cnt[ch * NB + bkt]; }
    boff[bkt + 1] = at;
  }
  ix->blocks.resize(boff[NB]);  // (not initialised: pass 2 writes every entry)
  ss_parallel_for(segs.size(), GRAIN, [&](size_t a, size_t b, unsigned) {
    uint64_t* at = start.data() + (a / GRAIN) * NB;
    for (size_t si = a; si < b; si++) {
      const Seg& sg = segs[si];
      const uint64_t head_bytes = sg.key_count * key_head_size;
      const uint8_t* body = bytes + sg.pos + head_bytes;
      const uint64_t body_len = sg.block_length - head_bytes;
      for (uint64_t i = 0; i < sg.key_count; i++) {
        const uint8_t* h = bytes + sg.pos + i * key_head_size;
        const uint64_t key = rd64(h);
        bool skip;
        const uint32_t n_comp = n_components(key, &skip);
        if (skip) continue;
        for (uint32_t c = 0; c < n_comp; c++) {
          ss_index_bin::Blk e{};
          e.key = key;
          e.n_comp = (uint8_t)n_comp;
          e.comp = (uint8_t)c;
          e.df_byte = (key & 7u) ? h[14 + c] : (uint8_t)0;
          e.b.block_id = sg.level;
          e.b.posting_count_m1 = (uint16_t)rd16(h + 8);
          e.b.pointer_pivot_p_docid = (uint16_t)rd16(h + key_head_size - 6);
          e.b.compression_type_pointer = rd32(h + key_head_size - 4);
          e.b.byte_array = body;
          e.b.byte_array_len = body_len;
          ix->blocks[at[key >> (64 - NB_LOG2)]++] = e;
        }
      }
    }
  });
  lap("key heads into the block table");
  // a bucket's terms while it is hot: runs of equal (key, component) -- a key never spans buckets
  struct TermPiece { std::vector<uint64_t> keys, first; std::vector<uint8_t> comp, ncomp, dfb; };
  std::vector<TermPiece> tp(NB);
  ss_parallel_for(NB, 1, [&](size_t a, size_t b, unsigned) {
    for (size_t bkt = a; bkt < b; bkt++) {
      ss_index_bin::Blk* dst = ix->blocks.data() + boff[bkt];
      const uint64_t at = boff[bkt + 1] - boff[bkt];
      std::sort(dst, dst + at, [](const ss_index_bin::Blk& x, const ss_index_bin::Blk& y) {
        return x.key != y.key ? x.key < y.key : x.comp != y.comp ? x.comp < y.comp : x.b.block_id < y.b.block_id;  // levels ascending
      });
      TermPiece& P = tp[bkt];
      for (uint64_t i = 0; i < at; i++) {
        const ss_index_bin::Blk& e = dst[i];
        if (i == 0 || e.key != dst[i - 1].key || e.comp != dst[i - 1].comp) {
          P.keys.push_back(e.key); P.comp.push_back(e.comp); P.ncomp.push_back(e.n_comp); P.dfb.push_back(e.df_byte);
          P.first.push_back(boff[bkt] + i);
        } else {
          P.dfb.back() = e.df_byte;  // df byte of the LAST level
        }
      }
    }
  });
  lap("buckets sorted, cut into terms");
  std::vector<uint64_t> toff(NB + 1, 0);
  for (unsigned bkt = 0; bkt < NB; bkt++) toff[bkt + 1] = toff[bkt] + tp[bkt].keys.size();
  const size_t n_terms = toff[NB];
  ix->keys.resize(n_terms); ix->term_comp.resize(n_terms); ix->term_ncomp.resize(n_terms); ix->term_df_byte.resize(n_terms);
  ix->term_block_off.resize(n_terms + 1);
  ss_parallel_for(NB, 1, [&](size_t a, size_t b, unsigned) {
    for (size_t bkt = a; bkt < b; bkt++) {
      const TermPiece& P = tp[bkt];
      const size_t n = P.keys.size();
      if (!n) continue;
      std::memcpy(ix->keys.data() + toff[bkt], P.keys.data(), n * 8);
      std::memcpy(ix->term_block_off.data() + toff[bkt], P.first.data(), n * 8);
      std::memcpy(ix->term_comp.data() + toff[bkt], P.comp.data(), n);
      std::memcpy(ix->term_ncomp.data() + toff[bkt], P.ncomp.data(), n);
      std::memcpy(ix->term_df_byte.data() + toff[bkt], P.dfb.data(), n);
    }
  });
  ix->term_block_off[n_terms] = ix->blocks.size();
  lap("terms");
  *out = ix.release();
  return SS_OK;
}

namespace {
// the terms of `order` (old term ids) become the handle's terms, in that order
void index_bin_reorder(ss_index_bin* ix, const std::vector<uint32_t>& order) {
  std::vector<uint64_t> off(order.size() + 1, 0);
  for (size_t i = 0; i < order.size(); i++) off[i + 1] = off[i] + (ix->term_block_off[order[i] + 1] - ix->term_block_off[order[i]]);
  std::vector<ss_index_bin::Blk, NoInitAlloc<ss_index_bin::Blk>> blocks(off[order.size()]);  // (filled by the workers below)
  std::vector<uint64_t> keys(order.size());
  std::vector<uint8_t> comp(order.size()), ncomp(order.size()), dfb(order.size());
  ss_parallel_for(order.size(), 4096, [&](size_t a, size_t b, unsigned) {
    for (size_t i = a; i < b; i++) {
      const uint32_t t = order[i];
      keys[i] = ix->keys[t]; comp[i] = ix->term_comp[t]; ncomp[i] = ix->term_ncomp[t]; dfb[i] = ix->term_df_byte[t];
      const uint64_t n = ix->term_block_off[t + 1] - ix->term_block_off[t];
      if (n) std::memcpy((void*)(blocks.data() + off[i]), (const void*)(ix->blocks.data() + ix->term_block_off[t]), n * sizeof(ss_index_bin::Blk));
    }
  });
  ix->blocks.swap(blocks);
  ix->keys.swap(keys);
  ix->term_comp.swap(comp); ix->term_ncomp.swap(ncomp); ix->term_df_byte.swap(dfb);
  ix->term_block_off.swap(off);
}
std::vector<uint64_t> index_bin_term_counts(const ss_index_bin* ix) {
  std::vector<uint64_t> n(ix->keys.size(), 0);
  ss_parallel_for(ix->keys.size(), 8192, [&](size_t a, size_t b, unsigned) {
    for (size_t t = a; t < b; t++) {
      uint64_t c = 0;
      for (uint64_t bi = ix->term_block_off[t]; bi < ix->term_block_off[t + 1]; bi++) c += (uint64_t)ix->blocks[bi].b.posting_count_m1 + 1u;
      n[t] = c;
    }
  });
  return n;
}
}  // namespace

// Keeps only the keys with at least min_posting_count postings (over all levels) -- the device image spends a directory
// row and a probe row per term and sub-block, which pays for the frequent terms that dominate query cost and not for the
// long tail of a real vocabulary; queries that touch a dropped term stay on the host's own path.
extern "C" int ss_index_bin_filter(ss_index_bin* ix, uint64_t min_posting_count, uint32_t* n_terms_kept) {
  if (!ix) return SS_EINVAL;
  const std::vector<uint64_t> n = index_bin_term_counts(ix);
  std::vector<uint32_t> order;
  for (size_t t = 0; t < ix->keys.size(); t++)
    if (n[t] >= min_posting_count) order.push_back((uint32_t)t);  // (the components of an n-gram key have the same count: kept or dropped together)
  index_bin_reorder(ix, order);
  if (n_terms_kept) *n_terms_kept = (uint32_t)ix->keys.size();
  return SS_OK;
}

// Two tiers instead of dropping the tail: the keys with at least dense_min_posting_count postings come first (ascending key
// hash) and become the DENSE image's terms, the rest follow (ascending key hash) and go to the SPARSE tier (bm25_sparse.hip: plain
// sorted lists, no directory / probe rows) -- a real vocabulary's millions of rare keys then cost 8 bytes per posting instead of
// a directory row each.  Term id = position in that order (ss_index_bin_term_keys); the host finds a key with two binary searches,
// one per tier (*n_dense_out = where the second starts).  The components of an n-gram key stay consecutive inside their tier.
extern "C" int ss_index_bin_tier(ss_index_bin* ix, uint64_t dense_min_posting_count, uint32_t* n_dense_out) {
  if (!ix) return SS_EINVAL;
  // (several indexed fields: the sparse tier takes the rare keys' merged lists -- an image whose boosts rule merged lists out refuses them)
  const std::vector<uint64_t> n = index_bin_term_counts(ix);
  std::vector<uint32_t> order;
  order.reserve(ix->keys.size());
  for (int tier = 0; tier < 2; tier++) {
    for (size_t t = 0; t < ix->keys.size(); t++)
      if ((n[t] >= dense_min_posting_count) == (tier == 0)) order.push_back((uint32_t)t);
    if (tier == 0) ix->n_dense = (uint32_t)order.size();
  }
  index_bin_reorder(ix, order);
  if (n_dense_out) *n_dense_out = ix->n_dense;
  return SS_OK;
}

extern "C" int ss_index_bin_close(ss_index_bin* ix) {
  delete ix;
  return SS_OK;
}

extern "C" int ss_index_bin_info(const ss_index_bin* ix, uint64_t* n_docs, uint64_t* positions_sum, uint32_t* n_levels,
                                 uint32_t* n_terms, uint32_t* n_ngram_keys_skipped) {
  if (!ix) return SS_EINVAL;
  if (n_docs) *n_docs = ix->n_docs;
  if (positions_sum) *positions_sum = ix->positions_sum;
  if (n_levels) *n_levels = (uint32_t)ix->doclen.size();
  if (n_terms) *n_terms = (uint32_t)ix->keys.size();
  if (n_ngram_keys_skipped) *n_ngram_keys_skipped = ix->n_ngram_keys;
  return SS_OK;
}

// Per term: components of its key (1 = SingleTerm), which component this term is, and for n-gram components the posting
// count of the component TERM, DOCUMENT_LENGTH_COMPRESSION[posting_count_ngram_i_compressed] (compress_postinglist.rs:105-106,
// 209-232) -- the df its idf is computed from (search.rs:3231-3262), not the length of the n-gram's own list.
extern "C" int ss_index_bin_term_ngram(const ss_index_bin* ix, uint8_t* n_components_out, uint8_t* component_out,
                                       uint32_t* component_df_out) {
  if (!ix) return SS_EINVAL;
  for (size_t t = 0; t < ix->keys.size(); t++) {
    if (n_components_out) n_components_out[t] = ix->term_ncomp[t];
    if (component_out) component_out[t] = ix->term_comp[t];
    if (component_df_out) component_df_out[t] = ix->term_ncomp[t] > 1 ? ss_byte4_to_int(ix->term_df_byte[t]) : 0u;
  }
  return SS_OK;
}

extern "C" int ss_index_bin_term_keys(const ss_index_bin* ix, uint64_t* keys_out) {
  if (!ix || !keys_out) return SS_EINVAL;
  std::memcpy(keys_out, ix->keys.data(), ix->keys.size() * sizeof(uint64_t));
  return SS_OK;
}

namespace {
// decoded postings of one term appended to docs / tfs
int index_bin_term(const ss_index_bin* ix, uint32_t term, std::vector<uint32_t>& docs, std::vector<uint16_t>& tfs,
                   uint16_t* d16, uint16_t* t16, std::vector<uint16_t>* pos = nullptr, std::vector<uint16_t>* npos = nullptr) {
  if (ix->n_fields != 1) return SS_ENOTSUP;  // BM25F field vectors: SURVEY section 8 f-2
  for (uint64_t bi = ix->term_block_off[term]; bi < ix->term_block_off[term + 1]; bi++) {
    const ss_ref_block& b = ix->blocks[bi].b;
    const int n = decode_block(&b, ix->blocks[bi].n_comp, ix->blocks[bi].comp, d16, t16, pos, npos);
    if (n < 0) return n;
    for (int i = 0; i < n; i++) {
      const uint64_t doc = ((uint64_t)b.block_id << 16) | d16[i];
      if (doc >= ix->n_docs) return SS_EINVAL;
      docs.push_back((uint32_t)doc);
      tfs.push_back(t16[i]);
    }
  }
  return SS_OK;
}
}  // namespace

namespace {
// decoded postings of the terms [t0, t1), terms in parallel: CSR offsets relative to the range, docs / tfs (+ positions and their
// per-posting counts).  A term's posting count is known from its key heads, so docs / tfs / counts are decoded straight into their
// final places; only the positions, whose number the heads do not tell, are gathered per chunk and copied once.  The arrays are
// NOT value-initialised (a vector would zero a gigabyte on one thread first): the worker that decodes a range touches its pages.
template <class T>
struct RawBuf {
  std::unique_ptr<T[]> p;
  size_t n = 0;
  void alloc(size_t m) { p.reset(new T[std::max<size_t>(m, 1)]); n = m; }
  T* data() { return p.get(); }
  const T* data() const { return p.get(); }
  size_t size() const { return n; }
};
struct DecodedRange {
  std::vector<uint64_t> offs;  // [t1 - t0 + 1]
  RawBuf<uint32_t> docs;
  RawBuf<uint16_t> tfs, pos, npos;
};
int index_bin_decode_range(const ss_index_bin* ix, uint32_t t0, uint32_t t1, bool with_positions, DecodedRange* out) {
  if (ix->n_fields != 1) return SS_ENOTSUP;
  const size_t nt = t1 - t0;
  static const bool trace = getenv("SS_LOAD_TRACE") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[load] decode: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
    t_last = t;
  };
  // chunks of terms with about the same number of postings each (a term's count is known from its key heads)
  std::vector<uint64_t>& cum = out->offs;
  cum.assign(nt + 1, 0);
  ss_parallel_for(nt, 16384, [&](size_t a, size_t b, unsigned) {
    for (size_t i = a; i < b; i++) {
      uint64_t c = 0;
      for (uint64_t bi = ix->term_block_off[t0 + i]; bi < ix->term_block_off[t0 + i + 1]; bi++) c += (uint64_t)ix->blocks[bi].b.posting_count_m1 + 1u;
      cum[i + 1] = c;
    }
  });
  for (size_t i = 0; i < nt; i++) cum[i + 1] += cum[i];
  const uint64_t total = cum[nt];
  const uint64_t per = std::max<uint64_t>(total / (8ull * ss_loader_threads()) + 1, 1u << 16);
  std::vector<size_t> cuts{0};
  for (size_t i = 1; i <= nt; i++)
    if (i == nt || cum[i] - cum[cuts.back()] >= per) cuts.push_back(i);
  const size_t nc = cuts.size() - 1;
  lap("posting counts, chunks");
  out->docs.alloc(total); out->tfs.alloc(total);
  if (with_positions) out->npos.alloc(total);
  std::vector<std::vector<uint16_t>> piece_pos(with_positions ? nc : 0);
  std::atomic<int> rc_all{SS_OK};
  ss_parallel_for(nc, 1, [&](size_t a, size_t b, unsigned) {
    std::vector<uint16_t> d16(65536), t16(65536), np;
    for (size_t c = a; c < b; c++) {
      std::vector<uint16_t>* pos = with_positions ? &piece_pos[c] : nullptr;
      if (pos) pos->reserve((size_t)((cum[cuts[c + 1]] - cum[cuts[c]]) * 5 / 4) + 1024);
      for (size_t i = cuts[c]; i < cuts[c + 1]; i++) {
        uint64_t at = cum[i];
        for (uint64_t bi = ix->term_block_off[t0 + i]; bi < ix->term_block_off[t0 + i + 1]; bi++) {
          const ss_ref_block& blk = ix->blocks[bi].b;
          np.clear();
          const int n = decode_block(&blk, ix->blocks[bi].n_comp, ix->blocks[bi].comp, d16.data(), t16.data(), pos, pos ? &np : nullptr);
          int rc = n < 0 ? n : SS_OK;
          if (rc == SS_OK && (at + (uint64_t)n > cum[i + 1] || (pos && np.size() != (size_t)n))) rc = SS_EINVAL;  // the key head's count is the block's
          if (rc == SS_OK && n && (((uint64_t)blk.block_id << 16) | d16[n - 1]) >= ix->n_docs) rc = SS_EINVAL;  // (ascending inside a block)
          if (rc) { int ok = SS_OK; rc_all.compare_exchange_strong(ok, rc); return; }
          uint32_t* dd = out->docs.data() + at;
          uint16_t* tt = out->tfs.data() + at;
          const uint32_t hi = (uint32_t)blk.block_id << 16;
          for (int j = 0; j < n; j++) { dd[j] = hi | d16[j]; tt[j] = t16[j]; }
          if (pos) std::memcpy(out->npos.data() + at, np.data(), (size_t)n * sizeof(uint16_t));
          at += (uint64_t)n;
        }
        if (at != cum[i + 1]) { int ok = SS_OK; rc_all.compare_exchange_strong(ok, SS_EINVAL); return; }
      }
    }
  });
  if (rc_all.load()) return rc_all.load();
  lap("terms decoded");
  if (with_positions) {
    std::vector<uint64_t> p_at(nc + 1, 0);
    for (size_t c = 0; c < nc; c++) p_at[c + 1] = p_at[c] + piece_pos[c].size();
    out->pos.alloc(p_at[nc]);
    ss_parallel_for(nc, 1, [&](size_t a, size_t b, unsigned) {
      for (size_t c = a; c < b; c++) {
        if (!piece_pos[c].empty()) std::memcpy(out->pos.data() + p_at[c], piece_pos[c].data(), piece_pos[c].size() * sizeof(uint16_t));
        std::vector<uint16_t>().swap(piece_pos[c]);
      }
    });
    lap("positions into place");
  }
  return SS_OK;
}
}  // namespace

namespace {
int decode_fields_range(const ss_index_bin* ix, uint32_t t0, uint32_t t1, std::vector<uint64_t>& offs, NoInitVec<uint32_t>& docs,
                        NoInitVec<uint8_t>& fields, NoInitVec<uint16_t>& tfs, NoInitVec<uint16_t>* pos, NoInitVec<uint16_t>* npos);
}
// Host only (no device): decodes every key of the index the way the uploads do -- on the loader's worker threads -- and reports what
// came out.  For a host that wants to know what an open will cost before it takes the shard's write lock, and for timing the decoder
// alone (tools/probes/decode_bench.py).
extern "C" int ss_index_bin_decode_stats(const ss_index_bin* ix, int with_positions, uint64_t* n_postings_out, uint64_t* n_positions_out) {
  if (!ix) return SS_EINVAL;
  uint64_t npost = 0, npos = 0;
  const uint32_t n_all = (uint32_t)ix->keys.size();
  if (ix->n_fields == 1) {
    DecodedRange D;
    const int rc = index_bin_decode_range(ix, 0, n_all, with_positions != 0, &D);
    if (rc) return rc;
    npost = D.docs.size(); npos = D.pos.size();
  } else {
    std::vector<uint64_t> offs;
    NoInitVec<uint32_t> docs;
    NoInitVec<uint8_t> fields;
    NoInitVec<uint16_t> tfs, pos, cnt;
    const int rc = decode_fields_range(ix, 0, n_all, offs, docs, fields, tfs, with_positions ? &pos : nullptr, with_positions ? &cnt : nullptr);
    if (rc) return rc;
    npost = docs.size(); npos = pos.size();
  }
  if (n_postings_out) *n_postings_out = npost;
  if (n_positions_out) *n_positions_out = npos;
  return SS_OK;
}

extern "C" int ss_index_bin_decode_all(const ss_index_bin* ix, uint64_t* offs_out, uint32_t* doc_ids_out, uint16_t* tfs_out, uint64_t postings_cap,
                                       uint16_t* npos_out, uint16_t* positions_out, uint64_t positions_cap) {
  if (!ix || !offs_out || !doc_ids_out || !tfs_out || (positions_out && !npos_out)) return SS_EINVAL;
  if (ix->n_fields != 1) return SS_ENOTSUP;
  DecodedRange D;
  const int rc = index_bin_decode_range(ix, 0, (uint32_t)ix->keys.size(), positions_out != nullptr, &D);
  if (rc) return rc;
  if (D.docs.size() > postings_cap || (positions_out && D.pos.size() > positions_cap)) return SS_EINVAL;
  std::memcpy(offs_out, D.offs.data(), D.offs.size() * sizeof(uint64_t));
  std::memcpy(doc_ids_out, D.docs.data(), D.docs.size() * sizeof(uint32_t));
  std::memcpy(tfs_out, D.tfs.data(), D.tfs.size() * sizeof(uint16_t));
  if (positions_out) {
    std::memcpy(npos_out, D.npos.data(), D.npos.size() * sizeof(uint16_t));
    std::memcpy(positions_out, D.pos.data(), D.pos.size() * sizeof(uint16_t));
  }
  return SS_OK;
}

extern "C" int ss_index_bin_term_postings(const ss_index_bin* ix, uint32_t term, uint64_t cap, uint32_t* docs_out,
                                          uint16_t* tfs_out, uint64_t* n_out) {
  if (!ix || term >= ix->keys.size() || !n_out) return SS_EINVAL;
  std::vector<uint32_t> docs;
  std::vector<uint16_t> tfs, d16(65536), t16(65536);
  const int rc = index_bin_term(ix, term, docs, tfs, d16.data(), t16.data());
  if (rc) return rc;
  *n_out = docs.size();
  if (docs.size() > cap) return SS_EINVAL;  // *n_out tells the needed capacity
  if (docs_out) std::memcpy(docs_out, docs.data(), docs.size() * 4);
  if (tfs_out) std::memcpy(tfs_out, tfs.data(), tfs.size() * 2);
  return SS_OK;
}

namespace {
// multi-field index: (doc, field, tf) entries of every term, doclen rearranged to [field][doc]
// the entries (doc, field, tf) of the keys [t0, t1), CSR over offs [t1 - t0 + 1]; the keys are decoded on the loader's worker threads in
// chunks of about equal posting counts (index_bin_decode_range's scheme), the pieces then copied into place
int decode_fields_range(const ss_index_bin* ix, uint32_t t0, uint32_t t1, std::vector<uint64_t>& offs, NoInitVec<uint32_t>& docs,
                        NoInitVec<uint8_t>& fields, NoInitVec<uint16_t>& tfs, NoInitVec<uint16_t>* pos, NoInitVec<uint16_t>* npos) {
  const uint32_t F = ix->n_fields;
  const size_t nt = t1 - t0;
  struct Piece { std::vector<uint32_t> docs; std::vector<uint8_t> fields; std::vector<uint16_t> tfs, pos, npos; std::vector<uint64_t> n_ent; size_t first = 0; };
  std::vector<uint64_t> cum(nt + 1, 0);
  ss_parallel_for(nt, 16384, [&](size_t a, size_t b, unsigned) {
    for (size_t i = a; i < b; i++) {
      uint64_t c = 0;
      for (uint64_t bi = ix->term_block_off[t0 + i]; bi < ix->term_block_off[t0 + i + 1]; bi++) c += (uint64_t)ix->blocks[bi].b.posting_count_m1 + 1u;
      cum[i + 1] = c;
    }
  });
  for (size_t i = 0; i < nt; i++) cum[i + 1] += cum[i];
  const uint64_t per = std::max<uint64_t>(cum[nt] / (8ull * ss_loader_threads()) + 1, 1u << 16);
  std::vector<size_t> cuts{0};
  for (size_t i = 1; i <= nt; i++)
    if (i == nt || cum[i] - cum[cuts.back()] >= per) cuts.push_back(i);
  const size_t nc = cuts.size() - 1;
  std::vector<Piece> pc(nc);
  std::atomic<int> rc_all{SS_OK};
  ss_parallel_for(nc, 1, [&](size_t ca, size_t cb, unsigned) {
    std::vector<uint16_t> d16(65536), t16((size_t)65536 * F);
    std::vector<uint32_t> first(65537);
    std::vector<uint8_t> f8((size_t)65536 * F);
    for (size_t c = ca; c < cb; c++) {
      Piece& P = pc[c];
      P.first = cuts[c];
      {  // entries >= postings (a doc may hold the term in several fields): room for a quarter more, grown from there
        const size_t room = (size_t)((cum[cuts[c + 1]] - cum[cuts[c]]) * 5 / 4) + 1024;
        P.docs.reserve(room); P.fields.reserve(room); P.tfs.reserve(room);
        if (npos) P.npos.reserve(room);
        if (pos) P.pos.reserve(room);
      }
      for (size_t i = cuts[c]; i < cuts[c + 1]; i++) {
        const uint32_t t = (uint32_t)(t0 + i);
        const size_t before = P.docs.size();
        for (uint64_t bi = ix->term_block_off[t]; bi < ix->term_block_off[t + 1]; bi++) {
          const ss_ref_block& b = ix->blocks[bi].b;
          const int n = decode_block_fields(&b, F, ix->longest_field_id, ix->blocks[bi].n_comp, ix->blocks[bi].comp, d16.data(), first.data(),
                                            f8.data(), t16.data(), pos ? &P.pos : nullptr, npos ? &P.npos : nullptr);
          if (n < 0) { int ok = SS_OK; rc_all.compare_exchange_strong(ok, n); return; }
          for (int x = 0; x < n; x++) {
            const uint64_t doc = ((uint64_t)b.block_id << 16) | d16[x];
            if (doc >= ix->n_docs) { int ok = SS_OK; rc_all.compare_exchange_strong(ok, (int)SS_EINVAL); return; }
            for (uint32_t e = first[x]; e < first[x + 1]; e++) {
              P.docs.push_back((uint32_t)doc);
              P.fields.push_back(f8[e]);
              P.tfs.push_back(t16[e]);
            }
          }
        }
        P.n_ent.push_back(P.docs.size() - before);
      }
    }
  });
  if (rc_all.load()) return rc_all.load();
  offs.assign(nt + 1, 0);
  std::vector<uint64_t> d_at(nc + 1, 0), p_at(nc + 1, 0);
  for (size_t c = 0; c < nc; c++) {
    d_at[c + 1] = d_at[c] + pc[c].docs.size();
    p_at[c + 1] = p_at[c] + pc[c].pos.size();
    uint64_t at = d_at[c];
    for (size_t x = 0; x < pc[c].n_ent.size(); x++) { offs[pc[c].first + x] = at; at += pc[c].n_ent[x]; }
  }
  offs[nt] = d_at[nc];
  docs.resize(d_at[nc]); fields.resize(d_at[nc]); tfs.resize(d_at[nc]);
  if (pos) pos->resize(p_at[nc]);
  if (npos) npos->resize(d_at[nc]);
  std::atomic<int> bad{0};
  ss_parallel_for(nc, 1, [&](size_t ca, size_t cb, unsigned) {
    for (size_t c = ca; c < cb; c++) {
      Piece& P = pc[c];
      if (!P.docs.empty()) {
        std::memcpy(docs.data() + d_at[c], P.docs.data(), P.docs.size() * 4);
        std::memcpy(fields.data() + d_at[c], P.fields.data(), P.fields.size());
        std::memcpy(tfs.data() + d_at[c], P.tfs.data(), P.tfs.size() * 2);
        if (npos) { if (P.npos.size() != P.docs.size()) bad.store(1); else std::memcpy(npos->data() + d_at[c], P.npos.data(), P.npos.size() * 2); }
      }
      if (pos && !P.pos.empty()) std::memcpy(pos->data() + p_at[c], P.pos.data(), P.pos.size() * 2);
      Piece().docs.swap(P.docs); Piece().pos.swap(P.pos);
    }
  });
  return bad.load() ? (int)SS_EINVAL : (int)SS_OK;
}

int upload_index_bin_fields(ss_shard* s, const ss_index_bin* ix, const float* boost, bool with_positions) {
  const uint32_t F = ix->n_fields;
  static const bool trace = getenv("SS_LOAD_TRACE") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t_0 = now();
  const uint32_t n_all = (uint32_t)ix->keys.size(), n_dense = std::min<uint32_t>(ix->n_dense, n_all);  // ss_index_bin_tier
  if (n_dense == 0) return SS_EINVAL;
  NoInitVec<uint16_t> pos, npos;
  std::vector<uint64_t> offs;
  NoInitVec<uint32_t> docs;
  NoInitVec<uint8_t> fields;
  NoInitVec<uint16_t> tfs;
  int rc = decode_fields_range(ix, 0, n_dense, offs, docs, fields, tfs, with_positions ? &pos : nullptr, with_positions ? &npos : nullptr);
  if (rc) return rc;
  const auto t_1 = now();
  std::vector<uint8_t> doclen((size_t)F * ix->n_docs);
  for (uint32_t f = 0; f < F; f++)
    for (size_t l = 0; l < ix->doclen.size(); l++) {
      const uint64_t d0 = (uint64_t)l << 16;
      if (d0 >= ix->n_docs) break;
      std::memcpy(doclen.data() + (size_t)f * ix->n_docs + d0, ix->doclen[l] + (size_t)f * 65536u,
                  (size_t)std::min<uint64_t>(65536u, ix->n_docs - d0));
    }
  if (with_positions)
    rc = ssi_bm25_upload_fields_positions(s, ix->n_docs, F, doclen.data(), boost, n_dense, offs.data(), docs.data(),
                                          fields.data(), tfs.data(), ix->positions_sum, pos.data(), pos.size(), npos.data());
  else
    rc = ssi_bm25_upload_fields(s, ix->n_docs, F, doclen.data(), boost, n_dense, offs.data(), docs.data(), fields.data(), tfs.data(),
                                ix->positions_sum);
  const auto t_2 = now();
  if (trace) fprintf(stderr, "[load] dense tier: decode %.0f ms (%zu entries, %zu positions), image build + upload %.0f ms\n", ms(t_0, t_1), docs.size(), pos.size(), ms(t_1, t_2));
  if (rc || n_dense == n_all) return rc;
  // the rare keys: their entries decoded the same way, their merged lists appended to the sparse tier (ids continue behind the dense ones)
  std::vector<uint64_t> r_offs;
  NoInitVec<uint32_t> r_docs;
  NoInitVec<uint8_t> r_fields;
  NoInitVec<uint16_t> r_tfs;
  NoInitVec<uint16_t> r_pos, r_npos;
  rc = decode_fields_range(ix, n_dense, n_all, r_offs, r_docs, r_fields, r_tfs, with_positions ? &r_pos : nullptr, with_positions ? &r_npos : nullptr);
  if (rc) return rc;
  const auto t_3 = now();
  if (with_positions) {
    rc = ss_bm25_append_sparse_fields_positions(s, n_all - n_dense, r_offs.data(), r_docs.data(), r_fields.data(), r_tfs.data(), r_pos.data(),
                                                r_pos.size(), r_npos.data(), nullptr);
    if (trace) fprintf(stderr, "[load] sparse tier: decode %.0f ms (%zu entries), append %.0f ms\n", ms(t_2, t_3), r_docs.size(), ms(t_3, now()));
    return rc;
  }
  if (false)
    return ss_bm25_append_sparse_fields_positions(s, n_all - n_dense, r_offs.data(), r_docs.data(), r_fields.data(), r_tfs.data(), r_pos.data(),
                                                  r_pos.size(), r_npos.data(), nullptr);
  return ss_bm25_append_sparse_fields(s, n_all - n_dense, r_offs.data(), r_docs.data(), r_fields.data(), r_tfs.data(), nullptr);
}
}  // namespace

extern "C" int ss_bm25_upload_index_bin_fields(ss_shard* s, const ss_index_bin* ix, const float* boost) {
  if (!s || !ix) return SS_EINVAL;
  if (ix->keys.empty() || ix->n_docs == 0) return SS_EINVAL;
  if (ix->n_fields < 2) return ss_bm25_upload_index_bin(s, ix);
  if (ix->n_fields > 8) return SS_ENOTSUP;
  return ss_guard([&] { return upload_index_bin_fields(s, ix, boost, false); }, SS_ENOMEM, SS_EDEVICE);  // (host-side decode buffers: std::bad_alloc -> SS_ENOMEM)
}

namespace {
int upload_index_bin_single(ss_shard* s, const ss_index_bin* ix, bool with_positions);
}
extern "C" int ss_bm25_upload_index_bin(ss_shard* s, const ss_index_bin* ix) {
  if (!s || !ix) return SS_EINVAL;
  if (ix->keys.empty() || ix->n_docs == 0) return SS_EINVAL;
  if (ix->n_fields > 1) return ss_bm25_upload_index_bin_fields(s, ix, nullptr);
  return ss_guard([&] { return upload_index_bin_single(s, ix, false); }, SS_ENOMEM, SS_EDEVICE);  // (host-side decode buffers: std::bad_alloc -> SS_ENOMEM)
}
// The image plus the positions of every posting, for phrase queries (SS_ENOTSUP for a position beyond 65 535).  An n-gram key's
// own positions go to its first component term (one indexed field; several: still SS_ENOTSUP).  Several indexed fields: boost = 1
// (ss_bm25_upload_index_bin_fields_positions takes boosts).
extern "C" int ss_bm25_upload_index_bin_positions(ss_shard* s, const ss_index_bin* ix) {
  if (!s || !ix) return SS_EINVAL;
  if (ix->keys.empty() || ix->n_docs == 0) return SS_EINVAL;
  if (ix->n_fields > 8) return SS_ENOTSUP;
  if (ix->n_fields > 1) return ss_guard([&] { return upload_index_bin_fields(s, ix, nullptr, true); }, SS_ENOMEM, SS_EDEVICE);
  return ss_guard([&] { return upload_index_bin_single(s, ix, true); }, SS_ENOMEM, SS_EDEVICE);  // (host-side decode buffers: std::bad_alloc -> SS_ENOMEM)
}
extern "C" int ss_bm25_upload_index_bin_fields_positions(ss_shard* s, const ss_index_bin* ix, const float* boost) {
  if (!s || !ix) return SS_EINVAL;
  if (ix->keys.empty() || ix->n_docs == 0) return SS_EINVAL;
  if (ix->n_fields < 2) return ss_bm25_upload_index_bin_positions(s, ix);
  if (ix->n_fields > 8) return SS_ENOTSUP;
  return ss_guard([&] { return upload_index_bin_fields(s, ix, boost, true); }, SS_ENOMEM, SS_EDEVICE);  // (host-side decode buffers: std::bad_alloc -> SS_ENOMEM)
}
namespace {
int upload_index_bin_single(ss_shard* s, const ss_index_bin* ix, bool with_positions) {
  const uint32_t n_all = (uint32_t)ix->keys.size(), n_dense = std::min<uint32_t>(ix->n_dense, n_all);  // ss_index_bin_tier
  if (n_dense == 0) return SS_EINVAL;           // the dense image needs at least one list
  // (positions with a sparse tier: both tiers get theirs -- a phrase naming a rare word is driven by that word's sparse list)
  DecodedRange D;
  int rc = index_bin_decode_range(ix, 0, n_dense, with_positions, &D);
  if (rc) return rc;
  std::vector<uint8_t> doclen(ix->doclen.size() * 65536u);
  for (size_t l = 0; l < ix->doclen.size(); l++) std::memcpy(doclen.data() + l * 65536u, ix->doclen[l], 65536u);  // field 0
  // avgdl = positions_sum_normalized / indexed_doc_count as the reference's reader computes it (index.rs:3480-3482)
  rc = ssi_bm25_upload(s, ix->n_docs, doclen.data(), n_dense, D.offs.data(), D.docs.data(), D.tfs.data(), ix->positions_sum);
  if (rc) return rc;
  if (with_positions) {
    rc = ssi_bm25_attach_positions(s, D.offs.data(), D.docs.data(), D.tfs.data(), D.pos.data(), D.pos.size(), D.npos.data());
    if (rc) return rc;
  }
  if (n_dense < n_all) {  // the rare keys: decoded the same way, appended to the sparse tier (term ids continue behind the dense ones)
    DecodedRange R;
    rc = index_bin_decode_range(ix, n_dense, n_all, with_positions, &R);
    if (rc) return rc;
    if (with_positions)  // (an n-gram key's own positions behind its first component, as in the dense tier)
      return ss_bm25_append_sparse_positions(s, n_all - n_dense, R.offs.data(), R.docs.data(), R.tfs.data(), R.pos.data(), R.pos.size(),
                                             R.npos.data(), nullptr);
    return ss_bm25_append_sparse(s, n_all - n_dense, R.offs.data(), R.docs.data(), R.tfs.data(), nullptr);
  }
  return rc;
}
}  // namespace
