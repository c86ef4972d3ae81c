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
// value is the count (read_singlefield_value, add_result.rs:2584-2606).  N-gram keys (extra tf values before the
// count) and multi-field postings are outside this reader: SS_ENOTSUP.
#include <cstring>
#include <vector>

#include "ss_common.h"

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

}  // namespace

// Decodes one block.  docs_out / tfs_out need room for 65 536 entries.  Returns the posting count or a negative code.
extern "C" int ss_ref_decode_block(const ss_ref_block* b, uint16_t* docs_out, uint16_t* tfs_out) {
  if (!b || !b->byte_array || !docs_out || !tfs_out) return SS_EINVAL;
  const uint8_t* a = b->byte_array;
  const uint64_t len = b->byte_array_len;
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
  for (uint32_t r = 0; r < count; r++) {
    uint32_t tf = 0;
    if (r < pivot) {
      const uint64_t at = range + (uint64_t)r * 2u;
      if (at + 2u > len) return SS_EINVAL;
      const uint32_t p = rd16(a + at);
      if (p & 0x8000u) {  // embedded: 10 -> one position, 11 -> two
        const uint32_t tag = p >> 14;
        tf = tag == 2u ? 1u : tag == 3u ? 2u : 0u;
      } else {
        const uint64_t back = p & 0x7FFFu;
        if (back > range || !read_vint(a, len, range - back, &tf)) return SS_EINVAL;
      }
    } else {
      const uint64_t at = range + (uint64_t)r * 3u - pivot;
      if (at + 3u > len) return SS_EINVAL;
      const uint32_t p = rd24(a + at);
      if (p & 0x800000u) {  // embedded: 100 / 101 / 110 / 111 -> 1..4 positions
        const uint32_t tag = p >> 21;
        tf = tag >= 4u ? tag - 3u : 0u;
      } else {
        const uint64_t back = p & 0x7FFFFFu;
        if (back > range || !read_vint(a, len, range - back, &tf)) return SS_EINVAL;
      }
    }
    if (tf == 0u || tf > 65535u) return SS_EINVAL;  // positions_count >= 1 always
    tfs_out[r] = (uint16_t)tf;
  }
  return (int)count;
}

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
