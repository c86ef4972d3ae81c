/*
 * ss_oracle.c -- CPU restatement of SeekStorm's query hot path (see ss_oracle.h header note).
 * TEST INFRASTRUCTURE ONLY; never linked into the product.  Plain C11, no dependencies.
 * Citations are file:line under /root/reference/seekstorm/src.
 */
#include "ss_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ SmallFloat */
/* index.rs:4237-4251 int_to_byte4 (NUM_FREE_VALUES = 24, index.rs:4232) */
uint8_t so_int_to_byte4(uint32_t i) {
  if (i < 24u) return (uint8_t)i;
  uint32_t ii = i - 24u;
  uint32_t num_bits = ii ? 32u - (uint32_t)__builtin_clz(ii) : 0u;
  if (num_bits < 4u) return (uint8_t)(24u + ii);
  uint32_t shift = num_bits - 4u;
  return (uint8_t)(24u + (((ii >> shift) & 0x07u) | ((shift + 1u) << 3)));
}
/* index.rs:4255-4268 byte4_to_int */
uint32_t so_byte4_to_int(uint8_t b) {
  if (b < 24u) return b;
  uint32_t i = (uint32_t)b - 24u, bits = i & 7u, shift = i >> 3;
  if (shift == 0) return 24u + bits;
  return 24u + ((bits | 8u) << (shift - 1u));
}
/* commit.rs:318-319: positions_sum_normalized as f32 / indexed_doc_count as f32 */
float so_avgdl(uint64_t positions_sum_normalized, uint64_t indexed_doc_count) {
  return (float)positions_sum_normalized / (float)indexed_doc_count;
}
/* commit.rs:321-325 */
void so_bm25_component_cache(float avgdl, float* out) {
  for (int i = 0; i < 256; i++) {
    float q = (float)so_byte4_to_int((uint8_t)i) / avgdl;
    out[i] = SO_K * (1.0f - SO_B + SO_B * q);
  }
}
/* search.rs:3225-3230: (((N - n + 0.5) / (n + 0.5)) + 1.0).ln(), all f32 */
float so_idf(uint64_t N, uint64_t n) {
  float Nf = (float)N, nf = (float)n;
  return logf(((Nf - nf + 0.5f) / (nf + 0.5f)) + 1.0f);
}
/* add_result.rs:1445-1447: idf * ((tf * (K + 1.0) / (tf + comp)) + SIGMA) */
float so_bm25_term(float idf, uint32_t tf, float comp) {
  float t = (float)tf;
  return idf * ((t * (SO_K + 1.0f) / (t + comp)) + SO_SIGMA);
}

/* ------------------------------------------------------------------ generator */
uint64_t so_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
uint64_t so_h(uint64_t seed, uint64_t a, uint64_t b) {
  return so_splitmix64(seed ^ (a * 0x9E3779B97F4A7C15ull) ^ (b * 0xC2B2AE3D27D4EB4Full));
}
void so_lex_doclen(uint64_t seed, uint64_t d0, uint64_t n, const uint8_t* tab, uint8_t* out) {
  for (uint64_t i = 0; i < n; i++) out[i] = tab[so_h(seed, 0, d0 + i) >> 54];
}
/* tf - 1 of a synthetic posting: geometric with p = 0.6 (tf = 1 + min(geom(p = 0.6), 254)), integer-exact so that the
 * device generator reproduces it bit for bit: P(j >= m) = 0.4^m, j = how many of floor(0.4^m * 2^32), m = 1.., lie above u */
uint32_t so_geom06(uint32_t u) {
  static const uint32_t T[24] = {1717986918u, 687194767u, 274877906u, 109951162u, 43980465u, 17592186u, 7036874u, 2814749u,
                                 1125899u,    450359u,    180143u,    72057u,     28823u,    11529u,    4611u,    1844u,
                                 737u,        295u,       118u,       47u,        18u,       7u,        3u,       1u};
  uint32_t j = 0;
  while (j < 24u && u < T[j]) j++;
  return j;
}
/* CLUSTERED corpora (seeds with bit 63 set): a term's density varies with the doc's cluster -- runs of 1024 (odd terms) or 8192 (even
 * terms) consecutive doc ids; in 70 % of its clusters the term is 8 times rarer than its threshold says, in 25 % as the threshold
 * says, in 5 % four times denser.  Doc ids of a list come in bursts (Rle / Bitmap-shaped blocks, uneven block maxima), as in a
 * corpus ordered by source or time; the uniform corpora (bit 63 clear) keep postings independent per doc. */
uint32_t so_lex_cluster_thresh(uint64_t seed, uint32_t term, uint64_t d, uint32_t thresh32) {
  const uint64_t c = (term & 1u) ? (d >> 10) : (d >> 13);
  const uint32_t r = (uint32_t)(so_h(seed ^ 0xC1ull, (uint64_t)term + 1u, c) >> 40) & 0xFFFFu;
  const uint64_t m = r < 45875u ? 1u : r < 62259u ? 8u : 32u;
  const uint64_t v = ((uint64_t)thresh32 * m) >> 3;
  return v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v;
}
uint64_t so_lex_term_postings(uint64_t seed, uint32_t term, uint32_t thresh32, uint64_t n_docs,
                              uint32_t* out_docs, uint16_t* out_tfs, uint64_t cap) {
  uint64_t c = 0;
  const int clustered = (int)(seed >> 63);
  for (uint64_t d = 0; d < n_docs; d++) {
    uint64_t hv = so_h(seed, (uint64_t)term + 1u, d);
    if ((uint32_t)(hv >> 32) < (clustered ? so_lex_cluster_thresh(seed, term, d, thresh32) : thresh32)) {
      if (out_docs && c < cap) {
        out_docs[c] = (uint32_t)d;
        out_tfs[c] = (uint16_t)(1u + so_geom06((uint32_t)hv));
      }
      c++;
    }
  }
  return c;
}
/* vector_similarity.rs:70-74 normalize_f32: norm = sqrt(sum x*x) (sequential), factor = 1/norm */
void so_normalize_f32(float* v, uint32_t dim) {
  volatile float s = 0.0f; /* volatile: forbid reassociation/FMA contraction, keep the reference's order */
  for (uint32_t i = 0; i < dim; i++) {
    volatile float p = v[i] * v[i];
    s = s + p;
  }
  float f = 1.0f / sqrtf(s);
  for (uint32_t i = 0; i < dim; i++) v[i] *= f;
}
void so_vec_gen(uint64_t seed, uint64_t r0, uint64_t n, uint32_t dim, int normalize, float* out) {
  for (uint64_t r = 0; r < n; r++) {
    float* row = out + r * dim;
    for (uint32_t c = 0; c < dim; c++) {
      int32_t iv = (int32_t)(uint32_t)(so_h(seed, r0 + r, c) >> 32);
      row[c] = (float)iv * 4.656612873077392578125e-10f; /* 2^-31 */
    }
    if (normalize) so_normalize_f32(row, dim);
  }
}

/* rows r0, r0 + stride, ...: the rows of ONE shard of a partitioned stream (row g -> shard g % S) without the others */
void so_vec_gen_strided(uint64_t seed, uint64_t r0, uint64_t stride, uint64_t n, uint32_t dim, int normalize, float* out) {
  for (uint64_t r = 0; r < n; r++) so_vec_gen(seed, r0 + r * stride, 1, dim, normalize, out + r * dim);
}

/* ------------------------------------------------------------------ shard model */
typedef struct {
  uint32_t block_id;
  uint32_t count; /* postings in block (reference stores count-1 as u16, index.rs:786) */
  int ctype;
  float max_part; /* idf-less block max (index.rs:2938 get_max_score, compress_postinglist.rs:529-555) */
  uint8_t* cont;  /* container bytes */
  uint32_t cont_bytes;
  const uint16_t* tf; /* decoded tf by rank (stands in for the position-pointer decode, add_result.rs:2036) */
} so_blk;
typedef struct {
  uint64_t posting_count;
  uint32_t n_blocks;
  so_blk* blocks;
} so_term;
struct so_shard {
  uint64_t n_docs;
  uint32_t n_level_blocks;
  uint8_t* doclen;
  float avgdl;
  float comp[256];
  uint32_t n_terms;
  so_term* terms;
  uint8_t* deleted; /* delete_hashset (index.rs:1594) as a byte per doc, NULL = empty */
  uint64_t* off;   /* raw CSR copy for the exhaustive ground truth */
  uint32_t* docs;
  uint16_t* tfs;
  uint64_t* pos_off; /* [postings + 1] CSR of the positions of every posting (so_shard_set_positions), NULL = none */
  uint16_t* pos;
};

static inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline void wr16(uint8_t* p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }

/* compress_postinglist.rs:256-332 chooser + 694 (array) / 759 (bitmap) / 832 (rle) writers */
static void encode_block(so_blk* b, const uint32_t* docs, uint32_t n) {
  uint32_t runs = 1;
  for (uint32_t i = 1; i < n; i++)
    if ((docs[i] & 0xFFFFu) != (docs[i - 1] & 0xFFFFu) + 1u) runs++;
  /* rle writer abandons when completed runs (= runs-1 at the end) reach the threshold:
   * threshold = count/2 if count < 4096 else 2048 (delta disabled, compress_postinglist.rs:242) */
  uint32_t thr = (n < 4096u) ? (n / 2u) : 2048u;
  int rle = (thr > 0u) && (runs - 1u < thr);
  if (rle) {
    b->ctype = SO_CT_RLE;
    b->cont_bytes = 2u + runs * 4u;
    b->cont = (uint8_t*)malloc(b->cont_bytes);
    wr16(b->cont, (uint16_t)runs);
    uint32_t r = 0, start = docs[0] & 0xFFFFu, len = 0;
    for (uint32_t i = 1; i < n; i++) {
      uint32_t d = docs[i] & 0xFFFFu;
      if (d == (docs[i - 1] & 0xFFFFu) + 1u) len++;
      else {
        wr16(b->cont + 2 + r * 4, (uint16_t)start);
        wr16(b->cont + 4 + r * 4, (uint16_t)len);
        r++; start = d; len = 0;
      }
    }
    wr16(b->cont + 2 + r * 4, (uint16_t)start);
    wr16(b->cont + 4 + r * 4, (uint16_t)len);
  } else if (n < 4096u) {
    b->ctype = SO_CT_ARRAY;
    b->cont_bytes = n * 2u;
    b->cont = (uint8_t*)malloc(b->cont_bytes);
    for (uint32_t i = 0; i < n; i++) wr16(b->cont + 2 * i, (uint16_t)(docs[i] & 0xFFFFu));
  } else {
    b->ctype = SO_CT_BITMAP;
    b->cont_bytes = 8192u;
    b->cont = (uint8_t*)calloc(8192u, 1);
    for (uint32_t i = 0; i < n; i++) {
      uint32_t d = docs[i] & 0xFFFFu; /* bit d <-> byte d>>3 bit d&7: compress_postinglist.rs:818-823 */
      b->cont[d >> 3] |= (uint8_t)(1u << (d & 7u));
    }
  }
}

static uint32_t decode_block(const so_blk* b, uint16_t* out) {
  uint32_t n = 0;
  if (b->ctype == SO_CT_ARRAY) {
    for (uint32_t i = 0; i < b->count; i++) out[n++] = rd16(b->cont + 2 * i);
  } else if (b->ctype == SO_CT_BITMAP) {
    for (uint32_t w = 0; w < 1024; w++) {
      uint64_t x;
      memcpy(&x, b->cont + 8 * w, 8);
      while (x) { /* intersection.rs:33-108 tzcnt / blsr iteration */
        out[n++] = (uint16_t)(w * 64u + (uint32_t)__builtin_ctzll(x));
        x &= x - 1;
      }
    }
  } else {
    uint32_t runs = rd16(b->cont);
    for (uint32_t r = 0; r < runs; r++) {
      uint32_t s = rd16(b->cont + 2 + 4 * r), l = rd16(b->cont + 4 + 4 * r);
      for (uint32_t j = 0; j <= l; j++) out[n++] = (uint16_t)(s + j); /* 0..=runlength, single.rs:235 */
    }
  }
  return n;
}

so_shard* so_shard_build(uint64_t n_docs, const uint8_t* doclen, uint32_t n_terms, const uint64_t* off,
                         const uint32_t* docs, const uint16_t* tfs) {
  so_shard* s = (so_shard*)calloc(1, sizeof(so_shard));
  s->n_docs = n_docs;
  s->n_level_blocks = (uint32_t)((n_docs + SO_BLOCK - 1) / SO_BLOCK);
  s->doclen = (uint8_t*)calloc((size_t)s->n_level_blocks * SO_BLOCK, 1);
  memcpy(s->doclen, doclen, n_docs);
  uint64_t psum = 0;
  for (uint64_t d = 0; d < n_docs; d++) psum += so_byte4_to_int(doclen[d]);
  s->avgdl = so_avgdl(psum, n_docs);
  so_bm25_component_cache(s->avgdl, s->comp);
  s->n_terms = n_terms;
  s->terms = (so_term*)calloc(n_terms, sizeof(so_term));
  uint64_t total = off[n_terms];
  s->off = (uint64_t*)malloc((n_terms + 1) * sizeof(uint64_t));
  memcpy(s->off, off, (n_terms + 1) * sizeof(uint64_t));
  s->docs = (uint32_t*)malloc((total ? total : 1) * sizeof(uint32_t));
  s->tfs = (uint16_t*)malloc((total ? total : 1) * sizeof(uint16_t));
  memcpy(s->docs, docs, total * sizeof(uint32_t));
  memcpy(s->tfs, tfs, total * sizeof(uint16_t));
  for (uint32_t t = 0; t < n_terms; t++) {
    so_term* T = &s->terms[t];
    uint64_t b0 = off[t], b1 = off[t + 1];
    T->posting_count = b1 - b0;
    uint32_t nb = 0;
    for (uint64_t i = b0; i < b1; i++)
      if (i == b0 || (docs[i] >> 16) != (docs[i - 1] >> 16)) nb++;
    T->n_blocks = nb;
    T->blocks = (so_blk*)calloc(nb ? nb : 1, sizeof(so_blk));
    uint32_t bi = 0;
    for (uint64_t i = b0; i < b1;) {
      uint64_t j = i;
      while (j < b1 && (docs[j] >> 16) == (docs[i] >> 16)) j++;
      so_blk* b = &T->blocks[bi++];
      b->block_id = docs[i] >> 16;
      b->count = (uint32_t)(j - i);
      b->tf = s->tfs + i;
      encode_block(b, docs + i, b->count);
      float mx = 0.0f;
      for (uint64_t p = i; p < j; p++) {
        float v = so_bm25_term(1.0f, tfs[p], s->comp[doclen[docs[p]]]);
        if (v > mx) mx = v;
      }
      b->max_part = mx;
      i = j;
    }
  }
  return s;
}
void so_shard_free(so_shard* s) {
  if (!s) return;
  for (uint32_t t = 0; t < s->n_terms; t++) {
    for (uint32_t b = 0; b < s->terms[t].n_blocks; b++) free(s->terms[t].blocks[b].cont);
    free(s->terms[t].blocks);
  }
  free(s->terms); free(s->doclen); free(s->off); free(s->docs); free(s->tfs); free(s->deleted); free(s->pos_off); free(s->pos); free(s);
}
/* delete_hashset: index.rs:1594, filled from delete.bin (index.rs:3798-3809) / delete_document (index.rs:5110) */
void so_shard_set_deleted(so_shard* s, const uint64_t* doc_ids, uint64_t n) {
  free(s->deleted);
  s->deleted = NULL;
  if (!n) return;
  s->deleted = (uint8_t*)calloc((size_t)s->n_level_blocks * SO_BLOCK, 1);
  for (uint64_t i = 0; i < n; i++) if (doc_ids[i] < (uint64_t)s->n_level_blocks * SO_BLOCK) s->deleted[doc_ids[i]] = 1;
}
float so_shard_avgdl(const so_shard* s) { return s->avgdl; }
uint64_t so_shard_posting_count(const so_shard* s, uint32_t t) { return t < s->n_terms ? s->terms[t].posting_count : 0; }
int so_shard_container(const so_shard* s, uint32_t t, uint32_t bo, uint32_t* bid, uint32_t* cnt, float* mp) {
  if (t >= s->n_terms || bo >= s->terms[t].n_blocks) return 0;
  const so_blk* b = &s->terms[t].blocks[bo];
  if (bid) *bid = b->block_id;
  if (cnt) *cnt = b->count;
  if (mp) *mp = b->max_part;
  return b->ctype;
}
uint32_t so_shard_decode_block(const so_shard* s, uint32_t t, uint32_t bo, uint16_t* out) {
  if (t >= s->n_terms || bo >= s->terms[t].n_blocks) return 0;
  return decode_block(&s->terms[t].blocks[bo], out);
}

/* ------------------------------------------------------------------ min-heap (min_heap.rs:45-53, 1193-1259) */
typedef struct { uint32_t doc; float score; } so_res;
typedef struct { so_res* e; uint32_t n, k; } so_heap;
static void heap_up(so_heap* h, uint32_t i) {
  while (i > 0) {
    uint32_t p = (i - 1) / 2;
    if (h->e[i].score < h->e[p].score) { so_res t = h->e[i]; h->e[i] = h->e[p]; h->e[p] = t; i = p; }
    else break;
  }
}
static void heap_down(so_heap* h, uint32_t i) {
  for (;;) {
    uint32_t l = 2 * i + 1, r = l + 1, m = i;
    if (l < h->n && h->e[l].score < h->e[m].score) m = l;
    if (r < h->n && h->e[r].score < h->e[m].score) m = r;
    if (m == i) break;
    so_res t = h->e[i]; h->e[i] = h->e[m]; h->e[m] = t; i = m;
  }
}
/* add_topk without docid_hashset: admit while not full, else only if score > root (STRICT, min_heap.rs:1250) */
static int heap_add_topk(so_heap* h, uint32_t doc, float score) {
  if (h->k == 0) return 0;
  if (h->n < h->k) { h->e[h->n].doc = doc; h->e[h->n].score = score; h->n++; heap_up(h, h->n - 1); return 1; }
  if (score > h->e[0].score) { h->e[0].doc = doc; h->e[0].score = score; heap_down(h, 0); return 1; }
  return 0;
}
static int heap_full(const so_heap* h) { return h->n >= h->k; }
/* search.rs:3565-3593: stable sort of the heap array by score desc */
static uint32_t heap_drain(so_heap* h, uint32_t* od, float* os) {
  for (uint32_t i = 1; i < h->n; i++) { /* insertion sort = stable */
    so_res x = h->e[i]; uint32_t j = i;
    while (j > 0 && h->e[j - 1].score < x.score) { h->e[j] = h->e[j - 1]; j--; }
    h->e[j] = x;
  }
  for (uint32_t i = 0; i < h->n; i++) { od[i] = h->e[i].doc; os[i] = h->e[i].score; }
  return h->n;
}

/* ------------------------------------------------------------------ container cursors (intersection.rs:112-2013) */
typedef struct {
  const so_blk* b;
  float idf;
  uint32_t pos;      /* array: index cursor; rle: run cursor */
  uint32_t rle_rank; /* rle: postings before run `pos` */
  uint32_t rank;     /* rank (p_docid) of the last successful lookup */
} so_cur;

/* membership + rank of doc id d (ascending probes).  Array: galloping, intersection.rs:352-362.
 * Bitmap: bit test + popcount rank, intersection.rs:772-794.  Rle: run walk, intersection.rs:934-. */
static int cur_find(so_cur* c, uint32_t d) {
  const so_blk* b = c->b;
  if (b->ctype == SO_CT_ARRAY) {
    uint32_t n = b->count, p = c->pos;
    if (p >= n) return 0;
    if (rd16(b->cont + 2 * p) < d) {
      uint32_t bound = 2;
      while (p + bound < n && rd16(b->cont + 2 * (p + bound)) < d) { p += bound; bound <<= 1; }
      uint32_t hi = p + bound < n ? p + bound : n - 1;
      uint32_t lo = p;
      while (lo < hi) { /* first index with value >= d in (p, hi] */
        uint32_t mid = (lo + hi) / 2;
        if (rd16(b->cont + 2 * mid) < d) lo = mid + 1; else hi = mid;
      }
      p = lo;
    }
    c->pos = p;
    if (p < n && rd16(b->cont + 2 * p) == d) { c->rank = p; return 1; }
    return 0;
  } else if (b->ctype == SO_CT_BITMAP) {
    if (!((b->cont[d >> 3] >> (d & 7u)) & 1u)) return 0;
    /* running popcount cursor (p_run = words summed, p_run_sum = their bits), intersection.rs:772-789: probes ascend */
    const uint32_t w = d >> 6;
    for (; c->pos < w; c->pos++) { uint64_t x; memcpy(&x, b->cont + 8 * c->pos, 8); c->rle_rank += (uint32_t)__builtin_popcountll(x); }
    uint64_t x; memcpy(&x, b->cont + 8 * w, 8);
    const uint32_t bit = d & 63u;
    c->rank = c->rle_rank + (bit ? (uint32_t)__builtin_popcountll(x & ((1ull << bit) - 1ull)) : 0u);
    return 1;
  } else {
    uint32_t runs = rd16(b->cont);
    while (c->pos < runs) {
      uint32_t s = rd16(b->cont + 2 + 4 * c->pos), l = rd16(b->cont + 4 + 4 * c->pos);
      if (d > s + l) { c->rle_rank += l + 1; c->pos++; continue; }
      if (d < s) return 0;
      c->rank = c->rle_rank + (d - s);
      return 1;
    }
    return 0;
  }
}

typedef struct { uint32_t ord[32]; float score; uint32_t block_id; uint32_t present; } so_bm;
static int bm_cmp(const void* a, const void* b) {
  const so_bm* x = (const so_bm*)a; const so_bm* y = (const so_bm*)b;
  if (x->score > y->score) return -1;
  if (x->score < y->score) return 1;
  return (x->block_id > y->block_id) - (x->block_id < y->block_id);
}

/* intersection_blockid (intersection.rs:2023-2301) + intersection_docid (112-447) +
 * add_result_multiterm_singlefield (add_result.rs:3418-3706, no filters / no phrase) */
static void search_and(const so_shard* s, uint32_t nq, const uint32_t* qt, const float* idf, int rt,
                       so_heap* heap, uint64_t* total, const uint8_t* gone) {
  uint32_t ptr[32] = {0};
  uint32_t nbm = 0, cap = 0;
  for (uint32_t t = 0; t < nq; t++) if (t == 0 || s->terms[qt[t]].n_blocks < cap) cap = s->terms[qt[t]].n_blocks;
  so_bm* bms = (so_bm*)malloc((cap ? cap : 1) * sizeof(so_bm));
  /* block-id merge, intersection.rs:2058-2222 */
  for (;;) {
    int done = 0; uint32_t mx = 0;
    for (uint32_t t = 0; t < nq; t++) {
      const so_term* T = &s->terms[qt[t]];
      if (ptr[t] >= T->n_blocks) { done = 1; break; }
      if (T->blocks[ptr[t]].block_id > mx) mx = T->blocks[ptr[t]].block_id;
    }
    if (done) break;
    int all = 1;
    for (uint32_t t = 0; t < nq; t++) {
      const so_term* T = &s->terms[qt[t]];
      while (ptr[t] < T->n_blocks && T->blocks[ptr[t]].block_id < mx) ptr[t]++;
      if (ptr[t] >= T->n_blocks) { done = 1; break; }
      if (T->blocks[ptr[t]].block_id != mx) all = 0;
    }
    if (done) break;
    if (!all) continue;
    so_bm* m = &bms[nbm++];
    m->block_id = mx; m->score = 0.0f;
    for (uint32_t t = 0; t < nq; t++) { /* block_score = sum max_block_score, intersection.rs:2090-2097 */
      m->ord[t] = ptr[t];
      m->score += idf[t] * s->terms[qt[t]].blocks[ptr[t]].max_part;
      ptr[t]++;
    }
  }
  if (rt != SO_RT_COUNT) qsort(bms, nbm, sizeof(so_bm), bm_cmp); /* intersection.rs:2225 */
  uint16_t* first = (uint16_t*)malloc(65536 * sizeof(uint16_t));
  for (uint32_t bi = 0; bi < nbm; bi++) {
    so_bm* m = &bms[bi];
    if (rt == SO_RT_TOPK && heap_full(heap) && heap->k > 0 && m->score <= heap->e[0].score) break; /* 2227-2233 */
    /* term order: non-bitmap first, then by block posting count asc (intersection.rs:258-273) */
    uint32_t order[32];
    for (uint32_t t = 0; t < nq; t++) order[t] = t;
    for (uint32_t i = 1; i < nq; i++) {
      uint32_t x = order[i]; uint32_t j = i;
      for (; j > 0; j--) {
        const so_blk* a = &s->terms[qt[order[j - 1]]].blocks[m->ord[order[j - 1]]];
        const so_blk* b = &s->terms[qt[x]].blocks[m->ord[x]];
        int abm = a->ctype == SO_CT_BITMAP, bbm = b->ctype == SO_CT_BITMAP;
        int gt = (abm != bbm) ? (abm > bbm) : (a->count > b->count);
        if (!gt) break;
        order[j] = order[j - 1];
      }
      order[j] = x;
    }
    so_cur cur[32];
    for (uint32_t i = 0; i < nq; i++) {
      uint32_t t = order[i];
      cur[i].b = &s->terms[qt[t]].blocks[m->ord[t]];
      cur[i].idf = idf[t]; cur[i].pos = 0; cur[i].rle_rank = 0; cur[i].rank = 0;
    }
    uint32_t n0 = decode_block(cur[0].b, first);
    for (uint32_t p0 = 0; p0 < n0; p0++) {
      uint32_t d = first[p0];
      int ok = 1;
      for (uint32_t i = 1; i < nq && ok; i++) ok = cur_find(&cur[i], d);
      if (!ok) continue;
      cur[0].rank = p0;
      uint32_t docid = (m->block_id << 16) | d;
      if (gone && gone[docid]) continue; /* add_result.rs:3435 delete_hashset, 3440-3497 not_query_list: first things add_result does */
      /* add_result.rs:3503-3537 */
      if (rt == SO_RT_COUNT) { (*total)++; continue; }
      if (heap_full(heap) && heap->k > 0 && m->score <= heap->e[0].score) {
        if (rt == SO_RT_TOPKCOUNT) (*total)++;
        continue;
      }
      float comp = s->comp[s->doclen[docid]];
      float bm25 = 0.0f; /* add_result.rs:1435-1449, terms in current query_list order */
      for (uint32_t i = 0; i < nq; i++) bm25 += so_bm25_term(cur[i].idf, cur[i].b->tf[cur[i].rank], comp);
      (*total)++;
      heap_add_topk(heap, docid, bm25);
    }
  }
  free(first); free(bms);
}

/* union_blockid / union_docid / union_scan_8|32 structure (union.rs:265, 32, 403-805).  The reference
 * answers 2..10-term OR by sub-query decomposition (union.rs:1168-1479) whose net result is the exact
 * top-k of the union under full BM25 over matched terms; this table scan is the
 * reference's own formulation of the same result (used there for >10 terms / counts). */
static void search_or(const so_shard* s, uint32_t nq, const uint32_t* qt, const float* idf, int rt,
                      so_heap* heap, uint64_t* total, const uint8_t* gone) {
  uint32_t ptr[32] = {0};
  uint32_t cap = 0, nbm = 0;
  for (uint32_t t = 0; t < nq; t++) cap += s->terms[qt[t]].n_blocks;
  so_bm* bms = (so_bm*)malloc((cap ? cap : 1) * sizeof(so_bm));
  for (;;) { /* union of block ids */
    uint32_t mn = 0xFFFFFFFFu;
    for (uint32_t t = 0; t < nq; t++) {
      const so_term* T = &s->terms[qt[t]];
      if (ptr[t] < T->n_blocks && T->blocks[ptr[t]].block_id < mn) mn = T->blocks[ptr[t]].block_id;
    }
    if (mn == 0xFFFFFFFFu) break;
    so_bm* m = &bms[nbm++];
    m->block_id = mn; m->score = 0.0f; m->present = 0;
    for (uint32_t t = 0; t < nq; t++) {
      const so_term* T = &s->terms[qt[t]];
      if (ptr[t] < T->n_blocks && T->blocks[ptr[t]].block_id == mn) {
        m->ord[t] = ptr[t]; m->present |= 1u << t;
        m->score += idf[t] * T->blocks[ptr[t]].max_part;
        ptr[t]++;
      }
    }
  }
  if (rt != SO_RT_COUNT) qsort(bms, nbm, sizeof(so_bm), bm_cmp);
  uint32_t* table = (uint32_t*)malloc(65536 * sizeof(uint32_t));
  uint16_t* tmp = (uint16_t*)malloc(65536 * sizeof(uint16_t));
  float* mstab = nq <= 10 ? (float*)malloc(((size_t)1 << nq) * sizeof(float)) : NULL;
  for (uint32_t bi = 0; bi < nbm; bi++) {
    so_bm* m = &bms[bi];
    int block_skip = heap_full(heap) && heap->k > 0 && m->score <= heap->e[0].score;
    if (rt == SO_RT_TOPK && block_skip) break;
    memset(table, 0, 65536 * sizeof(uint32_t));
    for (uint32_t t = 0; t < nq; t++) { /* union.rs:418-481 scatter term bit */
      if (!(m->present & (1u << t))) continue;
      uint32_t n = decode_block(&s->terms[qt[t]].blocks[m->ord[t]], tmp);
      for (uint32_t i = 0; i < n; i++) table[tmp[i]] |= 1u << t;
    }
    if (mstab) /* union.rs:538-546 */
      for (uint32_t i = 0; i < (1u << nq); i++) {
        float v = 0.0f;
        for (uint32_t j = 0; j < nq; j++)
          if ((i >> j) & 1u) v += (m->present & (1u << j)) ? idf[j] * s->terms[qt[j]].blocks[m->ord[j]].max_part : 0.0f;
        mstab[i] = v;
      }
    uint32_t rank[32] = {0};
    for (uint32_t d = 0; d < 65536; d++) { /* union.rs:553-592 */
      uint32_t bits = table[d];
      if (!bits) continue;
      /* union.rs:975-: union_count clears deleted docs; add_result.rs:3435 skips them (ranks still advance) */
      /* ... and union.rs:483-530 zeroes the docs of NOT terms in the scatter table */
      int out = gone && gone[((uint64_t)m->block_id << 16) | d];
      if (!out) (*total)++;
      if (!out && !block_skip && rt != SO_RT_COUNT) {
        float bound;
        if (mstab) bound = mstab[bits];
        else { bound = 0.0f; for (uint32_t j = 0; j < nq; j++) if ((bits >> j) & 1u) bound += idf[j] * s->terms[qt[j]].blocks[m->ord[j]].max_part; }
        if (!heap_full(heap) || bound > heap->e[0].score) {
          uint32_t docid = (m->block_id << 16) | d;
          float comp = s->comp[s->doclen[docid]];
          float bm25 = 0.0f;
          for (uint32_t j = 0; j < nq; j++) /* add_result.rs:1444-1446 skips terms with bm25_flag=false */
            if ((bits >> j) & 1u) bm25 += so_bm25_term(idf[j], s->terms[qt[j]].blocks[m->ord[j]].tf[rank[j]], comp);
          heap_add_topk(heap, docid, bm25);
        }
      }
      for (uint32_t j = 0; j < nq; j++) rank[j] += (bits >> j) & 1u;
    }
  }
  free(mstab); free(tmp); free(table); free(bms);
}

/* deleted docs + docs of the NOT terms as one byte map (NULL when there are none).  The reference walks the NOT lists
 * with per-term cursors inside add_result (add_result.rs:3440-3497); the result is the same set difference. */
static uint8_t* exclusion_map(const so_shard* s, uint32_t n_not, const uint32_t* not_terms) {
  if (!s->deleted && !n_not) return NULL;
  size_t n = (size_t)s->n_level_blocks * SO_BLOCK;
  uint8_t* g = (uint8_t*)calloc(n, 1);
  if (s->deleted) memcpy(g, s->deleted, n);
  for (uint32_t j = 0; j < n_not; j++)
    if (not_terms[j] < s->n_terms)
      for (uint64_t i = s->off[not_terms[j]]; i < s->off[not_terms[j] + 1]; i++) g[s->docs[i]] = 1;
  return g;
}
uint32_t so_search_lex(const so_shard* s, uint32_t nq, const uint32_t* qt, int op, uint32_t k, int rt,
                       uint32_t* od, float* os, uint64_t* total) {
  return so_search_lex_not(s, nq, qt, 0, NULL, op, k, rt, od, os, total);
}
uint32_t so_search_lex_not(const so_shard* s, uint32_t nq, const uint32_t* qt, uint32_t n_not, const uint32_t* not_terms,
                           int op, uint32_t k, int rt, uint32_t* od, float* os, uint64_t* total) {
  uint64_t tot = 0;
  if (nq == 0 || nq > 32) { if (total) *total = 0; return 0; }
  float idf[32];
  for (uint32_t t = 0; t < nq; t++) {
    if (qt[t] >= s->n_terms) { if (total) *total = 0; return 0; }
    idf[t] = so_idf(s->n_docs, s->terms[qt[t]].posting_count);
  }
  /* search.rs:2527-2541: heap size min(offset+length, indexed_doc_count) */
  uint32_t kk = k; if ((uint64_t)kk > s->n_docs) kk = (uint32_t)s->n_docs;
  if (rt == SO_RT_COUNT) kk = 0;
  so_heap heap; heap.n = 0; heap.k = kk; heap.e = (so_res*)malloc((kk ? kk : 1) * sizeof(so_res));
  uint8_t* gone = exclusion_map(s, n_not, not_terms);
  if (op == SO_OP_AND || nq == 1) search_and(s, nq, qt, idf, rt, &heap, &tot, gone);
  else search_or(s, nq, qt, idf, rt, &heap, &tot, gone);
  uint32_t n = heap_drain(&heap, od, os);
  free(gone);
  free(heap.e);
  if (total) *total = tot;
  return n;
}

typedef struct { float score; uint32_t doc; } so_sd;
static int sd_cmp(const void* a, const void* b) {
  const so_sd* x = (const so_sd*)a; const so_sd* y = (const so_sd*)b;
  if (x->score > y->score) return -1;
  if (x->score < y->score) return 1;
  return (x->doc > y->doc) - (x->doc < y->doc);
}
uint32_t so_search_lex_exhaustive(const so_shard* s, uint32_t nq, const uint32_t* qt, int op, uint32_t k,
                                  uint32_t* od, float* os, uint64_t* total) {
  return so_search_lex_exhaustive_not(s, nq, qt, 0, NULL, op, k, od, os, total);
}
uint32_t so_search_lex_exhaustive_not(const so_shard* s, uint32_t nq, const uint32_t* qt, uint32_t n_not,
                                      const uint32_t* not_terms, int op, uint32_t k, uint32_t* od, float* os,
                                      uint64_t* total) {
  return so_search_lex_exhaustive_idf(s, nq, qt, NULL, n_not, not_terms, op, k, od, os, total);
}
/* idf_in != NULL: the idf of every query term is given.  An n-gram key is searched as its component terms -- posting lists
 * with the same docs and the component's tf -- each with idf_ngram_i from the component TERM's posting count
 * (search.rs:3231-3262): their sum is the n-gram arm of get_bm25f_multiterm_singlefield (add_result.rs:1454-1477). */
uint32_t so_search_lex_exhaustive_idf(const so_shard* s, uint32_t nq, const uint32_t* qt, const float* idf_in, uint32_t n_not,
                                      const uint32_t* not_terms, int op, uint32_t k, uint32_t* od, float* os,
                                      uint64_t* total) {
  return so_search_lex_exhaustive_opt(s, nq, qt, idf_in, n_not, not_terms, op, k, 0, od, os, total);
}
/* all_terms_frequent (intersection.rs:198-209): indexed_doc_count > top_k << 8 and posting_count / indexed_doc_count >= 0.5
 * (f32) for every term of an intersection */
int so_all_terms_frequent(const so_shard* s, uint32_t nq, const uint32_t* qt, uint32_t top_k) {
  if (!(s->n_docs > ((uint64_t)top_k << 8))) return 0;
  for (uint32_t t = 0; t < nq; t++)
    if ((float)s->terms[qt[t]].posting_count / (float)s->n_docs < 0.5f) return 0;
  return 1;
}
/* shortcut != 0: an intersection of several terms is run as the reference runs it under all_terms_frequent (the caller
 * decides with so_all_terms_frequent): a doc in which some term has an embedded pointer (<= 4 positions) or fewer than 10
 * positions is counted but not scored (add_result.rs:2091-2104, 3541-3556) -- ranked only if every tf >= 10 */
uint32_t so_search_lex_exhaustive_opt(const so_shard* s, uint32_t nq, const uint32_t* qt, const float* idf_in, uint32_t n_not,
                                      const uint32_t* not_terms, int op, uint32_t k, int shortcut, uint32_t* od, float* os,
                                      uint64_t* total) {
  float* sc = (float*)calloc(s->n_docs ? s->n_docs : 1, sizeof(float));
  uint8_t* cnt = (uint8_t*)calloc(s->n_docs ? s->n_docs : 1, 1);
  uint8_t* low = (uint8_t*)calloc(s->n_docs ? s->n_docs : 1, 1);  /* some term with tf < 10 */
  shortcut = shortcut && op == SO_OP_AND && nq > 1;
  for (uint32_t t = 0; t < nq; t++) {
    float idf = idf_in ? idf_in[t] : so_idf(s->n_docs, s->terms[qt[t]].posting_count);
    for (uint64_t i = s->off[qt[t]]; i < s->off[qt[t] + 1]; i++) {
      uint32_t d = s->docs[i];
      sc[d] += so_bm25_term(idf, s->tfs[i], s->comp[s->doclen[d]]);
      cnt[d]++;
      if (s->tfs[i] < 10) low[d] = 1;
    }
  }
  uint8_t* gone = exclusion_map(s, n_not, not_terms);
  if (gone)
    for (uint64_t d = 0; d < s->n_docs; d++) if (gone[d]) cnt[d] = 0;
  free(gone);
  uint64_t m = 0;
  for (uint64_t d = 0; d < s->n_docs; d++) if (op == SO_OP_AND ? cnt[d] == nq : cnt[d] > 0) m++;
  so_sd* v = (so_sd*)malloc((m ? m : 1) * sizeof(so_sd));
  uint64_t j = 0;
  for (uint64_t d = 0; d < s->n_docs; d++)
    if ((op == SO_OP_AND ? cnt[d] == nq : cnt[d] > 0) && !(shortcut && low[d])) { v[j].score = sc[d]; v[j].doc = (uint32_t)d; j++; }
  qsort(v, j, sizeof(so_sd), sd_cmp);
  uint32_t n = (uint32_t)(j < k ? j : k);
  for (uint32_t i = 0; i < n; i++) { od[i] = v[i].doc; os[i] = v[i].score; }
  if (total) *total = m;
  free(v); free(cnt); free(sc); free(low);
  return n;
}
void so_query_stats(const so_shard* s, uint32_t nq, const uint32_t* qt, uint64_t* sum_df, uint64_t* sum_blocks) {
  uint64_t a = 0, b = 0;
  for (uint32_t t = 0; t < nq; t++) { a += s->terms[qt[t]].posting_count; b += s->terms[qt[t]].n_blocks; }
  if (sum_df) *sum_df = a;
  if (sum_blocks) *sum_blocks = b;
}

/* ------------------------------------------------------------------ vector path */
/* vector_similarity.rs:1006-1008: a.iter().zip(b).map(|(x,y)| x*y).sum()  (sequential) */
float so_dot_f32(const float* a, const float* b, uint32_t dim) {
  float s = 0.0f;
  for (uint32_t i = 0; i < dim; i++) s += a[i] * b[i];
  return s;
}
/* vector_similarity.rs:1118-1142: 8 fmadd lanes over dim/8 steps, then the 8 lanes summed in order */
float so_dot_f32_lanes8(const float* q, const float* e, uint32_t dim) {
  float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t steps = dim / 8;
  for (uint32_t i = 0; i < steps; i++)
    for (int j = 0; j < 8; j++) l[j] = __builtin_fmaf(q[i * 8 + j], e[i * 8 + j], l[j]);
  float s = 0.0f;
  for (int j = 0; j < 8; j++) s += l[j];
  return s;
}
float so_vector_score_field(float dot) { return ((dot * (1.0f / 16129.0f)) + 1.0f) * 0.5f; }
/* vector.rs:388-397 (Dot/Cosine arm): ((t*2)-1) / SIMILARITY_NORMALIZATION_64_I8 */
float so_threshold_raw(float t) { return ((t * 2.0f) - 1.0f) / (1.0f / 16129.0f); }

typedef struct { uint32_t doc; float score; } so_item;
uint32_t so_vec_search(const float* rows, uint64_t n_rows, uint32_t dim, const uint32_t* row_doc, const float* q,
                       uint32_t k, float thr, int simd_order, uint32_t* od, float* os, uint64_t* out_total,
                       uint64_t* out_observed) {
  return so_vec_search_del(rows, n_rows, dim, row_doc, q, k, thr, simd_order, NULL, 0, od, os, out_total, out_observed);
}
/* search_vector_shard, AnnMode::All (vector.rs:1397-1466) over any record scorer: TopK::new / push, vector.rs:366-496 */
typedef float (*so_score_fn)(const void* ctx, uint64_t row);
typedef struct { so_item* items; uint32_t len, k; float thr, lowest; uint64_t total; } so_topk;
static void topk_new(so_topk* t, uint32_t k, float thr) {
  t->items = (so_item*)malloc((k ? k : 1) * sizeof(so_item));
  for (uint32_t i = 0; i < k; i++) { t->items[i].doc = 0; t->items[i].score = -FLT_MAX; }
  t->len = 0; t->k = k; t->thr = thr; t->lowest = -FLT_MAX; t->total = 0;
}
/* TopK::push, vector.rs:410-496 */
static void topk_push(so_topk* t, uint32_t doc, float score) {
  so_item* items = t->items;
  const uint32_t len = t->len, k = t->k;
  if (score < t->thr || (len == k && score <= t->lowest)) return;
  t->total++;
  if (len < k) {
    for (uint32_t i = 0; i < len; i++) if (items[i].doc == doc) { if (score > items[i].score) items[i].score = score; return; }
    items[len].doc = doc; items[len].score = score; t->len++;
    return;
  }
  uint32_t min_i = 0; float min_v = items[0].score;
  for (uint32_t i = 0; i < len; i++) {
    if (items[i].doc == doc) { if (score > items[i].score) items[i].score = score; return; }
    if (items[i].score < min_v) { min_v = items[i].score; min_i = i; }
  }
  if (score > min_v) { t->lowest = min_v; items[min_i].doc = doc; items[min_i].score = score; }
}
static void topk_sort_desc(so_topk* t) { /* vector.rs:1472 sort desc (stable) */
  so_item* items = t->items;
  for (uint32_t i = 1; i < t->len; i++) {
    so_item x = items[i]; uint32_t j = i;
    while (j > 0 && items[j - 1].score < x.score) { items[j] = items[j - 1]; j--; }
    items[j] = x;
  }
}
/* order != NULL: the rows visited, in visiting order (the ANN modes); NULL: rows 0 .. n_rows-1 */
static uint32_t topk_scan_order(uint64_t n_rows, const uint64_t* order, const uint32_t* row_doc, uint32_t k, float thr,
                                const uint64_t* deleted_sorted, uint64_t n_deleted, so_score_fn fn, const void* ctx, uint32_t* od,
                                float* os, uint64_t* out_total, uint64_t* out_observed) {
  so_topk t; topk_new(&t, k, thr);
  uint64_t observed = 0;
  for (uint64_t i = 0; i < n_rows; i++) {
    const uint64_t r = order ? order[i] : i;
    float score = fn(ctx, r);
    uint32_t doc = row_doc ? row_doc[r] : (uint32_t)r;
    if (n_deleted) { /* vector.rs:1450-1452: scored, then not pushed */
      uint64_t lo = 0, hi = n_deleted;
      while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (deleted_sorted[mid] < doc) lo = mid + 1; else hi = mid; }
      if (lo < n_deleted && deleted_sorted[lo] == doc) continue;
    }
    observed++; /* observed_vector_count is counted INSIDE TopK::push (vector.rs:421): a tombstoned record never gets there */
    if (k == 0) continue;
    topk_push(&t, doc, score);
  }
  topk_sort_desc(&t);
  for (uint32_t i = 0; i < t.len; i++) { od[i] = t.items[i].doc; os[i] = t.items[i].score; }
  if (out_total) *out_total = t.total;
  if (out_observed) *out_observed = observed;
  uint32_t len = t.len;
  free(t.items);
  return len;
}
static uint32_t topk_scan(uint64_t n_rows, const uint32_t* row_doc, uint32_t k, float thr, const uint64_t* deleted_sorted,
                          uint64_t n_deleted, so_score_fn fn, const void* ctx, uint32_t* od, float* os, uint64_t* out_total,
                          uint64_t* out_observed) {
  return topk_scan_order(n_rows, NULL, row_doc, k, thr, deleted_sorted, n_deleted, fn, ctx, od, os, out_total, out_observed);
}
/* The ANN modes of search_vector_shard (vector.rs:1300-1392): per level, every cluster's medoid (= its first record) is
 * scored and pushed into TopK::new(min(n_probe, clusters), cluster_threshold); the survivors are sorted by score desc
 * (stable over the TopK array) and their records visited cluster after cluster.  Returns the number of rows in order[]
 * (caller-allocated, >= total rows); *n_clusters_visited = observed_cluster_count. */
static uint64_t ann_order(uint32_t n_levels, const uint32_t* level_clusters, const uint32_t* child_count, uint32_t n_probe,
                          float cluster_thr, so_score_fn fn, const void* ctx, uint64_t* order, uint64_t* n_clusters_visited) {
  uint64_t n = 0, row0 = 0, visited = 0;
  const uint32_t* cc = child_count;
  for (uint32_t l = 0; l < n_levels; l++) {
    const uint32_t C = level_clusters[l];
    uint64_t* start = (uint64_t*)malloc((C ? C : 1) * sizeof(uint64_t));
    uint64_t level_rows = 0;
    for (uint32_t c = 0; c < C; c++) { start[c] = row0 + level_rows; level_rows += cc[c]; }
    so_topk t; topk_new(&t, n_probe < C ? n_probe : C, cluster_thr);
    if (t.k) for (uint32_t c = 0; c < C; c++) topk_push(&t, c, fn(ctx, start[c]));
    topk_sort_desc(&t);
    visited += t.len;
    for (uint32_t i = 0; i < t.len; i++) {
      const uint32_t c = t.items[i].doc;
      for (uint32_t j = 0; j < cc[c]; j++) order[n++] = start[c] + j;
    }
    free(t.items); free(start);
    row0 += level_rows; cc += C;
  }
  if (n_clusters_visited) *n_clusters_visited = visited;
  return n;
}

/* rows visited by search_vector_shard, in order: every cluster (AnnMode::All, n_levels = 0: rows 0..n-1) or ann_order's;
 * with a field filter only the records of the listed indexed fields (vector.rs:1397-1400: the others are skipped before
 * they are scored -- neither observed nor pushed; medoids are scored whatever their field) */
static uint64_t visit_order(uint64_t n_rows, uint32_t n_levels, const uint32_t* level_clusters, const uint32_t* child_count,
                            uint32_t n_probe, float cluster_thr, so_score_fn fn, const void* ctx, const uint16_t* row_field,
                            uint64_t field_mask, uint64_t* order, uint64_t* n_clusters_visited) {
  uint64_t n;
  if (n_levels == 0) {
    for (n = 0; n < n_rows; n++) order[n] = n;
    if (n_clusters_visited) *n_clusters_visited = 0;
  } else {
    n = ann_order(n_levels, level_clusters, child_count, n_probe, cluster_thr, fn, ctx, order, n_clusters_visited);
  }
  if (row_field && field_mask) {
    uint64_t w = 0;
    for (uint64_t i = 0; i < n; i++) {
      const uint16_t f = row_field[order[i]];
      if (f < 64 && ((field_mask >> f) & 1ull)) order[w++] = order[i];
    }
    n = w;
  }
  return n;
}

/* euclidean_f32 (vector_similarity.rs:912-918): sum of (x - y).powi(2), sequential; euclidean_f32_avx2 (938-966): eight
 * lanes of sub / mul / add over dim / 8 steps, then the lanes summed 0..7.  The similarity is MINUS this (arm at 337 / 907):
 * larger is closer, TopK unchanged. */
float so_euclidean_f32(const float* a, const float* b, uint32_t dim) {
  float s = 0.0f;
  for (uint32_t i = 0; i < dim; i++) { float d = a[i] - b[i]; s += d * d; }
  return s;
}
float so_euclidean_f32_lanes8(const float* q, const float* e, uint32_t dim) {
  float lane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t i = 0; i + 8 <= dim; i += 8)
    for (uint32_t l = 0; l < 8; l++) { float d = q[i + l] - e[i + l]; float sq = d * d; lane[l] = lane[l] + sq; }
  float s = 0.0f;
  for (uint32_t l = 0; l < 8; l++) s += lane[l];
  return s;
}
typedef struct { const float* rows; const float* q; uint32_t dim; int simd; int euclid; } so_f32_ctx;
static float score_f32(const void* c, uint64_t r) {
  const so_f32_ctx* x = (const so_f32_ctx*)c;
  const float* e = x->rows + r * x->dim;
  if (x->euclid) return -(x->simd ? so_euclidean_f32_lanes8(x->q, e, x->dim) : so_euclidean_f32(x->q, e, x->dim));
  return x->simd ? so_dot_f32_lanes8(x->q, e, x->dim) : so_dot_f32(x->q, e, x->dim);
}
uint32_t so_vec_search_del(const float* rows, uint64_t n_rows, uint32_t dim, const uint32_t* row_doc, const float* q,
                           uint32_t k, float thr, int simd_order, const uint64_t* deleted_sorted, uint64_t n_deleted,
                           uint32_t* od, float* os, uint64_t* out_total, uint64_t* out_observed) {
  so_f32_ctx c = {rows, q, dim, simd_order, 0};
  return topk_scan(n_rows, row_doc, k, thr, deleted_sorted, n_deleted, score_f32, &c, od, os, out_total, out_observed);
}

/* ---- i8 embeddings: quantize_f32_to_i8 (vector_similarity.rs:1226-1232), dot_i8 (1011-1016), dot_i8_quantized (1754-1758) */
void so_quantize_f32_to_i8(const float* v, uint32_t n, int8_t* out) {
  for (uint32_t i = 0; i < n; i++) {
    float x = roundf(v[i] * 127.0f); /* f32::round: half away from zero */
    if (x < -127.0f) x = -127.0f;
    if (x > 127.0f) x = 127.0f;
    out[i] = (int8_t)x;
  }
}
int32_t so_dot_i8(const int8_t* a, const int8_t* b, uint32_t dim) {
  int32_t s = 0;
  for (uint32_t i = 0; i < dim; i++) s += (int32_t)a[i] * (int32_t)b[i];
  return s;
}
typedef struct { const int8_t* rows; const int8_t* q; uint32_t dim; const float* row_scale; int scaled; float q_scale;
                 int euclid; const float* row_norm; float q_norm; } so_i8_ctx;
static float score_i8(const void* c, uint64_t r) {
  const so_i8_ctx* x = (const so_i8_ctx*)c;
  const int8_t* e = x->rows + r * x->dim;
  if (x->euclid && !x->scaled) { /* -euclidean_i8 (vector_similarity.rs:921-932): integer sum of squared differences as f32 */
    int32_t s = 0;
    for (uint32_t i = 0; i < x->dim; i++) { int32_t d = (int32_t)x->q[i] - (int32_t)e[i]; s += d * d; }
    return -(float)s;
  }
  int32_t d = so_dot_i8(x->q, e, x->dim);
  if (x->euclid) { /* -euclidean_i8_quantized (1721-1735): (norm1 + norm2 - 2.0 * dot).max(0.0), dot = dot_i32 as f32 * scale1 * scale2 */
    float dot = (float)d * x->q_scale * (x->row_scale ? x->row_scale[r] : 1.0f);
    float v = x->q_norm + (x->row_norm ? x->row_norm[r] : 0.0f) - 2.0f * dot;
    return -(v > 0.0f ? v : 0.0f);
  }
  if (!x->scaled) return (float)d;                                   /* dot_i8(a, b) as f32 */
  return (float)d * x->q_scale * (x->row_scale ? x->row_scale[r] : 1.0f); /* dot_i32 as f32 * scale1 * scale2 */
}
uint32_t so_vec_search_i8(const int8_t* rows, uint64_t n_rows, uint32_t dim, const uint32_t* row_doc, const float* row_scale,
                          const int8_t* q, int scaled, float q_scale, uint32_t k, float thr, const uint64_t* deleted_sorted,
                          uint64_t n_deleted, uint32_t* od, float* os, uint64_t* out_total, uint64_t* out_observed) {
  so_i8_ctx c = {rows, q, dim, row_scale, scaled, q_scale, 0, NULL, 0.0f};
  return topk_scan(n_rows, row_doc, k, thr, deleted_sorted, n_deleted, score_i8, &c, od, os, out_total, out_observed);
}

uint32_t so_vec_search_ann(const float* rows, uint64_t n_rows, uint32_t dim, const uint32_t* row_doc, const float* q,
                           uint32_t k, float thr, int simd_order, uint32_t n_levels, const uint32_t* level_clusters,
                           const uint32_t* child_count, uint32_t n_probe, float cluster_thr, const uint64_t* deleted_sorted,
                           uint64_t n_deleted, const uint16_t* row_field, uint64_t field_mask, uint32_t* od, float* os,
                           uint64_t* out_total, uint64_t* out_observed, uint64_t* out_clusters) {
  so_f32_ctx c = {rows, q, dim, simd_order, 0};
  uint64_t* order = (uint64_t*)malloc((n_rows ? n_rows : 1) * sizeof(uint64_t));
  uint64_t n = visit_order(n_rows, n_levels, level_clusters, child_count, n_probe, cluster_thr, score_f32, &c, row_field,
                           field_mask, order, out_clusters);
  uint32_t r = topk_scan_order(n, order, row_doc, k, thr, deleted_sorted, n_deleted, score_f32, &c, od, os, out_total, out_observed);
  free(order);
  return r;
}
uint32_t so_vec_search_i8_ann(const int8_t* rows, uint64_t n_rows, uint32_t dim, const uint32_t* row_doc, const float* row_scale,
                              const int8_t* q, int scaled, float q_scale, uint32_t k, float thr, uint32_t n_levels,
                              const uint32_t* level_clusters, const uint32_t* child_count, uint32_t n_probe, float cluster_thr,
                              const uint64_t* deleted_sorted, uint64_t n_deleted, const uint16_t* row_field,
                              uint64_t field_mask, uint32_t* od, float* os, uint64_t* out_total, uint64_t* out_observed,
                              uint64_t* out_clusters) {
  so_i8_ctx c = {rows, q, dim, row_scale, scaled, q_scale, 0, NULL, 0.0f};
  uint64_t* order = (uint64_t*)malloc((n_rows ? n_rows : 1) * sizeof(uint64_t));
  uint64_t n = visit_order(n_rows, n_levels, level_clusters, child_count, n_probe, cluster_thr, score_i8, &c, row_field,
                           field_mask, order, out_clusters);
  uint32_t r = topk_scan_order(n, order, row_doc, k, thr, deleted_sorted, n_deleted, score_i8, &c, od, os, out_total, out_observed);
  free(order);
  return r;
}

/* VectorSimilarity::Euclidean (vector_similarity.rs:257-345, 905-907): the similarity of a record is MINUS the squared
 * distance, so the same TopK / thresholds apply (threshold_raw = -similarity_threshold, vector.rs:398).  n_levels = 0:
 * AnnMode::All.  f32: euclidean_f32 (sequential) or euclidean_f32_avx2 lane order.  i8: quantized != 0 ->
 * euclidean_i8_quantized with (scale, norm) of query and records, else euclidean_i8 (exact integer). */
uint32_t so_vec_search_euclid(const float* rows, uint64_t n_rows, uint32_t dim, const uint32_t* row_doc, const float* q,
                              uint32_t k, float thr, int simd_order, uint32_t n_levels, const uint32_t* level_clusters,
                              const uint32_t* child_count, uint32_t n_probe, float cluster_thr, const uint64_t* deleted_sorted,
                              uint64_t n_deleted, const uint16_t* row_field, uint64_t field_mask, uint32_t* od, float* os,
                              uint64_t* out_total, uint64_t* out_observed, uint64_t* out_clusters) {
  so_f32_ctx c = {rows, q, dim, simd_order, 1};
  uint64_t* order = (uint64_t*)malloc((n_rows ? n_rows : 1) * sizeof(uint64_t));
  uint64_t n = visit_order(n_rows, n_levels, level_clusters, child_count, n_probe, cluster_thr, score_f32, &c, row_field,
                           field_mask, order, out_clusters);
  uint32_t r = topk_scan_order(n, order, row_doc, k, thr, deleted_sorted, n_deleted, score_f32, &c, od, os, out_total, out_observed);
  free(order);
  return r;
}
uint32_t so_vec_search_i8_euclid(const int8_t* rows, uint64_t n_rows, uint32_t dim, const uint32_t* row_doc, const float* row_scale,
                                 const float* row_norm, const int8_t* q, int quantized, float q_scale, float q_norm, uint32_t k,
                                 float thr, uint32_t n_levels, const uint32_t* level_clusters, const uint32_t* child_count,
                                 uint32_t n_probe, float cluster_thr, const uint64_t* deleted_sorted, uint64_t n_deleted,
                                 const uint16_t* row_field, uint64_t field_mask, uint32_t* od, float* os, uint64_t* out_total,
                                 uint64_t* out_observed, uint64_t* out_clusters) {
  so_i8_ctx c = {rows, q, dim, row_scale, quantized, q_scale, 1, row_norm, q_norm};
  uint64_t* order = (uint64_t*)malloc((n_rows ? n_rows : 1) * sizeof(uint64_t));
  uint64_t n = visit_order(n_rows, n_levels, level_clusters, child_count, n_probe, cluster_thr, score_i8, &c, row_field,
                           field_mask, order, out_clusters);
  uint32_t r = topk_scan_order(n, order, row_doc, k, thr, deleted_sorted, n_deleted, score_i8, &c, od, os, out_total, out_observed);
  free(order);
  return r;
}

/* ------------------------------------------------------------------ merge / RRF */
typedef struct { uint64_t doc; float score; uint8_t src; uint32_t seq; } so_m;
static int m_cmp_desc_stable(const void* a, const void* b) {
  const so_m* x = (const so_m*)a; const so_m* y = (const so_m*)b;
  if (x->score > y->score) return -1;
  if (x->score < y->score) return 1;
  return (x->seq > y->seq) - (x->seq < y->seq);
}
static int m_cmp_desc_doc(const void* a, const void* b) {
  const so_m* x = (const so_m*)a; const so_m* y = (const so_m*)b;
  if (x->score > y->score) return -1;
  if (x->score < y->score) return 1;
  return (x->doc > y->doc) - (x->doc < y->doc);
}
uint32_t so_merge(int mode, const uint64_t* ld, const float* ls, uint32_t nl, const uint64_t* vd, const float* vs,
                  uint32_t nv, uint32_t offset, uint32_t length, uint64_t* od, float* os, uint8_t* osrc) {
  uint32_t cap = nl + nv + 1, n = 0;
  so_m* out = (so_m*)malloc(cap * sizeof(so_m));
  if (mode == 0) { for (uint32_t i = 0; i < nl; i++) { out[n].doc = ld[i]; out[n].score = ls[i]; out[n].src = 0; out[n].seq = n; n++; } }
  else if (mode == 1) { for (uint32_t i = 0; i < nv; i++) { out[n].doc = vd[i]; out[n].score = vs[i]; out[n].src = 1; out[n].seq = n; n++; } }
  else { /* search.rs:1962-2035: k = 0.6, 0-based rank over the cross-shard concatenation sorted desc */
    so_m* L = (so_m*)malloc((nl + 1) * sizeof(so_m));
    so_m* V = (so_m*)malloc((nv + 1) * sizeof(so_m));
    for (uint32_t i = 0; i < nl; i++) { L[i].doc = ld[i]; L[i].score = ls[i]; L[i].seq = i; }
    for (uint32_t i = 0; i < nv; i++) { V[i].doc = vd[i]; V[i].score = vs[i]; V[i].seq = i; }
    qsort(L, nl, sizeof(so_m), m_cmp_desc_stable);
    qsort(V, nv, sizeof(so_m), m_cmp_desc_stable);
    const float kk = 0.6f;
    for (uint32_t i = 0; i < nl; i++) { /* insert: a later duplicate doc id overwrites (AHashMap::insert) */
      uint32_t j = 0; for (; j < n; j++) if (out[j].doc == L[i].doc) break;
      out[j].doc = L[i].doc; out[j].score = 1.0f / (kk + (float)i); out[j].src = 0; out[j].seq = j;
      if (j == n) n++;
    }
    for (uint32_t i = 0; i < nv; i++) {
      float r = 1.0f / (kk + (float)i);
      uint32_t j = 0; for (; j < n; j++) if (out[j].doc == V[i].doc) break;
      if (j < n) { out[j].score += r; out[j].src = 2; }
      else { out[n].doc = V[i].doc; out[n].score = r; out[n].src = 1; out[n].seq = n; n++; }
    }
    free(L); free(V);
  }
  /* search.rs:2103-2105 stable sort by score desc; lexical/vector keep concatenation order on ties,
   * hybrid order on ties is hash order in the reference -> doc id asc here (deterministic) */
  qsort(out, n, sizeof(so_m), mode == 2 ? m_cmp_desc_doc : m_cmp_desc_stable);
  uint32_t w = 0;
  for (uint32_t i = offset; i < n && w < length; i++, w++) { od[w] = out[i].doc; os[w] = out[i].score; if (osrc) osrc[w] = out[i].src; }
  free(out);
  return w;
}

/* ------------------------------------------------------------------ several indexed fields (BM25F)
 * get_bm25f_multiterm_multifield (add_result.rs:1171-1426), SingleTerm keys: for every query term present in the doc and
 * every field it occurs in:  bm25f += boost[field] * idf * (tf (K+1) / (tf + comp[len_byte(doc, field)]) + SIGMA),
 * fields in ascending order inside a term, terms in query order.  comp = the ONE bm25_component_cache of the shard
 * (avgdl = positions_sum over all fields / indexed_doc_count, commit.rs:318-325).  idf from the docs containing the term in
 * any field (posting_count).  Intersection: every term in at least one field.  Brute force, exact top-k by (score desc,
 * doc asc): the ground truth of the multi-field parity tests.  Entries of a term sorted by (doc, field). */
uint32_t so_search_fields_exhaustive(uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen /*[n_fields][n_docs]*/,
                                     const float* boost, const uint64_t* off, const uint32_t* docs, const uint8_t* fields,
                                     const uint16_t* tfs, uint32_t nq, const uint32_t* qt, uint32_t n_not,
                                     const uint32_t* not_terms, int op, uint32_t k, const uint64_t* deleted, uint64_t n_deleted,
                                     uint32_t* od, float* os, uint64_t* total, float* out_avgdl) {
  return so_search_fields_filtered(n_docs, n_fields, doclen, boost, off, docs, fields, tfs, nq, qt, n_not, not_terms, op, k, deleted,
                                   n_deleted, 0u, od, os, total, out_avgdl);
}
/* field_mask != 0: the query's field_filter (add_result.rs:3124-3136, add_result_multiterm_multifield): a doc is dropped
 * unless EVERY query term it is matched on occurs in at least one listed field (bit f = indexed field f); the score still
 * sums every field.  Restated for intersections and single-term queries, where "the terms it is matched on" is the whole
 * query; a union of several terms reaches add_result through union_docid_3's sub-queries (union.rs:1308-1479), whose
 * interplay with the filter is not a function of the doc alone -- not modelled. */
static uint32_t fields_core(uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen, const float* boost, const uint64_t* off,
                            const uint32_t* docs, const uint8_t* fields, const uint16_t* tfs, uint32_t nq, const uint32_t* qt,
                            uint32_t n_not, const uint32_t* not_terms, int op, uint32_t k, const uint64_t* deleted, uint64_t n_deleted,
                            uint32_t field_mask, int shortcut, uint32_t* od, float* os, uint64_t* total, float* out_avgdl);
uint32_t so_search_fields_filtered(uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen /*[n_fields][n_docs]*/,
                                   const float* boost, const uint64_t* off, const uint32_t* docs, const uint8_t* fields,
                                   const uint16_t* tfs, uint32_t nq, const uint32_t* qt, uint32_t n_not,
                                   const uint32_t* not_terms, int op, uint32_t k, const uint64_t* deleted, uint64_t n_deleted,
                                   uint32_t field_mask, uint32_t* od, float* os, uint64_t* total, float* out_avgdl) {
  return fields_core(n_docs, n_fields, doclen, boost, off, docs, fields, tfs, nq, qt, n_not, not_terms, op, k, deleted, n_deleted,
                     field_mask, 0, od, os, total, out_avgdl);
}
/* an INTERSECTION under the all_terms_frequent shortcut, several indexed fields (decode_positions_multiterm_multifield returns
 * true = "count the doc, do not rank it", add_result.rs:1595-1607, 3111-3122): for an embedded pointer always, for a record when its
 * FIRST field has fewer than 10 positions.  A posting whose first field holds >= 10 positions is never embedded (index_posting.rs:
 * 437: embedding stops at 4 positions), so: a doc is ranked only if, for every query term, the term's positions count in the LOWEST
 * field that holds the doc is >= 10; every matching doc is counted.  The caller has checked the condition
 * (so_all_terms_frequent's: N > 256 k, every df >= N / 2); no field filter (3116). */
uint32_t so_search_fields_shortcut(uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen, const float* boost, const uint64_t* off,
                                   const uint32_t* docs, const uint8_t* fields, const uint16_t* tfs, uint32_t nq, const uint32_t* qt,
                                   uint32_t k, const uint64_t* deleted, uint64_t n_deleted, uint32_t* od, float* os, uint64_t* total) {
  return fields_core(n_docs, n_fields, doclen, boost, off, docs, fields, tfs, nq, qt, 0, NULL, SO_OP_AND, k, deleted, n_deleted, 0u, 1,
                     od, os, total, NULL);
}
static uint32_t fields_core(uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen, const float* boost, const uint64_t* off,
                            const uint32_t* docs, const uint8_t* fields, const uint16_t* tfs, uint32_t nq, const uint32_t* qt,
                            uint32_t n_not, const uint32_t* not_terms, int op, uint32_t k, const uint64_t* deleted, uint64_t n_deleted,
                            uint32_t field_mask, int shortcut, uint32_t* od, float* os, uint64_t* total, float* out_avgdl) {
  uint8_t* low = shortcut ? (uint8_t*)calloc(n_docs ? n_docs : 1, 1) : NULL;  /* some term's first field has < 10 positions */
  uint64_t psum = 0;
  for (uint64_t i = 0; i < n_docs * n_fields; i++) psum += so_byte4_to_int(doclen[i]);
  const float avgdl = so_avgdl(psum, n_docs);
  if (out_avgdl) *out_avgdl = avgdl;
  float comp[256];
  so_bm25_component_cache(avgdl, comp);
  float* sc = (float*)calloc(n_docs ? n_docs : 1, sizeof(float));
  uint8_t* cnt = (uint8_t*)calloc(n_docs ? n_docs : 1, 1);
  for (uint32_t t = 0; t < nq; t++) {
    uint64_t df = 0;
    for (uint64_t i = off[qt[t]]; i < off[qt[t] + 1]; i++) if (i == off[qt[t]] || docs[i] != docs[i - 1]) df++;
    const float idf = so_idf(n_docs, df);
    for (uint64_t i = off[qt[t]]; i < off[qt[t] + 1]; i++) {
      const uint32_t d = docs[i];
      const float w = boost ? boost[fields[i]] : 1.0f;
      sc[d] += w * idf * ((float)tfs[i] * (SO_K + 1.0f) / ((float)tfs[i] + comp[doclen[(uint64_t)fields[i] * n_docs + d]]) + SO_SIGMA);
      if (i == off[qt[t]] || docs[i] != docs[i - 1]) {  // first entry of (term, doc): does the term pass the filter here?
        if (low && tfs[i] < 10) low[d] = 1;             // ... = the doc's lowest field for this term
        int hit = field_mask == 0;
        for (uint64_t j = i; j < off[qt[t] + 1] && docs[j] == d && !hit; j++) hit = (field_mask >> fields[j]) & 1u;
        if (hit) cnt[d]++;
      }
    }
  }
  for (uint32_t j = 0; j < n_not; j++)
    for (uint64_t i = off[not_terms[j]]; i < off[not_terms[j] + 1]; i++) cnt[docs[i]] = 0xFF;
  for (uint64_t i = 0; i < n_deleted; i++) if (deleted[i] < n_docs) cnt[deleted[i]] = 0xFF;
  uint64_t m = 0;
  for (uint64_t d = 0; d < n_docs; d++) if (cnt[d] != 0xFF && (op == SO_OP_AND ? cnt[d] == nq : cnt[d] > 0)) m++;
  so_sd* v = (so_sd*)malloc((m ? m : 1) * sizeof(so_sd));
  uint64_t j = 0;
  for (uint64_t d = 0; d < n_docs; d++)
    if (cnt[d] != 0xFF && (op == SO_OP_AND ? cnt[d] == nq : cnt[d] > 0) && !(low && low[d])) { v[j].score = sc[d]; v[j].doc = (uint32_t)d; j++; }
  qsort(v, j, sizeof(so_sd), sd_cmp);
  uint32_t n = (uint32_t)(j < k ? j : k);
  for (uint32_t i = 0; i < n; i++) { od[i] = v[i].doc; os[i] = v[i].score; }
  if (total) *total = m;
  free(v); free(cnt); free(sc); free(low);
  return n;
}

/* ================================================================== reference-structured dispatch
 * The dispatch block of search_lexical_shard (search.rs:3374-3560) as the reference runs it for ONE indexed field:
 *   1 term                      -> single_blockid                       (single.rs:292-417, single_docid 23-289)
 *   union of 2                  -> union_docid_2                        (union.rs:1168-1305)
 *   union of 3..10, Topk[Count] -> union_docid_3 (sub-query queue)      (union.rs:1308-1479)
 *   union Count / > 10 terms    -> union_blockid -> union_scan          (search_or above)
 *   intersection                -> intersection_blockid                 (search_and above)
 * with MinHeap::add_topk's docid_hashset arm (min_heap.rs:1193-1260), which is what lets the sub-queries of a union
 * re-offer a doc that an earlier sub-query already placed.  This is the path the CPU baseline times (bench.py).  */
typedef struct { uint32_t* key; float* val; uint32_t cap, n, used; } so_hset; /* docid_hashset: doc -> score, min_heap.rs:46 */
#define HS_EMPTY 0xFFFFFFFFu
#define HS_TOMB 0xFFFFFFFEu
static void hset_init(so_hset* h, uint32_t cap) {
  h->cap = cap; h->n = 0; h->used = 0;
  h->key = (uint32_t*)malloc(cap * sizeof(uint32_t)); h->val = (float*)malloc(cap * sizeof(float));
  memset(h->key, 0xFF, cap * sizeof(uint32_t));
}
static void hset_free(so_hset* h) { free(h->key); free(h->val); }
static inline uint32_t hs_slot(const so_hset* h, uint32_t d) { return (d * 2654435761u) & (h->cap - 1); }
static int hset_get(const so_hset* h, uint32_t d, float* v) {
  for (uint32_t i = hs_slot(h, d);; i = (i + 1) & (h->cap - 1)) {
    if (h->key[i] == HS_EMPTY) return 0;
    if (h->key[i] == d) { *v = h->val[i]; return 1; }
  }
}
static void hset_put(so_hset* h, uint32_t d, float v);
static void hset_grow(so_hset* h) {
  so_hset o = *h;
  hset_init(h, o.cap * 2);
  for (uint32_t i = 0; i < o.cap; i++) if (o.key[i] < HS_TOMB) hset_put(h, o.key[i], o.val[i]);
  hset_free(&o);
}
static void hset_put(so_hset* h, uint32_t d, float v) { /* HashMap::insert: overwrites */
  if ((h->used + 1) * 2 > h->cap) hset_grow(h);
  int32_t tomb = -1;
  for (uint32_t i = hs_slot(h, d);; i = (i + 1) & (h->cap - 1)) {
    if (h->key[i] == d) { h->val[i] = v; return; }
    if (h->key[i] == HS_TOMB && tomb < 0) tomb = (int32_t)i;
    if (h->key[i] == HS_EMPTY) {
      if (tomb >= 0) i = (uint32_t)tomb; else h->used++;
      h->key[i] = d; h->val[i] = v; h->n++;
      return;
    }
  }
}
static void hset_remove(so_hset* h, uint32_t d) {
  for (uint32_t i = hs_slot(h, d);; i = (i + 1) & (h->cap - 1)) {
    if (h->key[i] == HS_EMPTY) return;
    if (h->key[i] == d) { h->key[i] = HS_TOMB; h->n--; return; }
  }
}

typedef struct { so_heap h; so_hset hs; } so_topk_ref;
/* pop_add, min_heap.rs:1113-1121 */
static void ref_pop_add(so_topk_ref* T, uint32_t doc, float score) {
  if (T->hs.n) hset_remove(&T->hs, T->h.e[0].doc);
  T->h.e[0].doc = doc; T->h.e[0].score = score;
  heap_down(&T->h, 0);
}
/* add_topk, min_heap.rs:1193-1260 (result_sort empty: ordering = score) */
static int ref_add_topk(so_topk_ref* T, uint32_t doc, float score) {
  so_heap* h = &T->h;
  if (h->k == 0) return 0;
  float old;
  if (T->hs.n && hset_get(&T->hs, doc, &old)) {
    if (h->e[0].doc == doc) {
      if (score > h->e[0].score) { h->e[0].score = score; heap_down(h, 0); return 1; }
      return 0;
    }
    if (old >= score) return 0;
    uint32_t idx = 0;
    while (h->e[idx].doc != doc) {
      if (idx == h->n - 1) { ref_pop_add(T, doc, score); return 1; }
      idx++;
    }
    h->e[idx].score = score;
    heap_down(h, idx);
    return 1;
  }
  if (h->n < h->k) { h->e[h->n].doc = doc; h->e[h->n].score = score; h->n++; heap_up(h, h->n - 1); return 1; }
  if (score > h->e[0].score) { ref_pop_add(T, doc, score); return 1; }
  return 0;
}
/* "for i in 0..current_heap_size: docid_hashset.insert(doc, score)" before a sub-query, union.rs:1254-1259, 1348-1353, 1437-1442 */
static void ref_snapshot(so_topk_ref* T) {
  for (uint32_t i = 0; i < T->h.n; i++) hset_put(&T->hs, T->h.e[i].doc, T->h.e[i].score);
}

typedef struct { float score; uint32_t ord; } so_sb;
static int sb_cmp(const void* a, const void* b) {
  const so_sb* x = (const so_sb*)a; const so_sb* y = (const so_sb*)b;
  if (x->score > y->score) return -1;
  if (x->score < y->score) return 1;
  return (x->ord > y->ord) - (x->ord < y->ord);
}
/* single_blockid + single_docid + add_result_singleterm_singlefield (single.rs:292-417, 23-289; add_result.rs:647-900).
 * `filtered` as single.rs:307-312 (NOT terms or a delete set).  Blocks in block_score order, at most top_k of them when
 * unfiltered, stop / skip at block_score <= heap minimum.  Returns what the function adds to result_count. */
static uint64_t ref_single(const so_shard* s, uint32_t term, float idf, int rt, uint32_t top_k, so_topk_ref* T,
                           const uint8_t* gone, int unique_terms) {
  const so_term* Tm = &s->terms[term];
  const int filtered = gone != NULL;
  if (rt == SO_RT_COUNT && unique_terms <= 1 && !filtered) return Tm->posting_count; /* single.rs:314-324 */
  uint64_t local = 0;
  so_sb* bv = (so_sb*)malloc((Tm->n_blocks ? Tm->n_blocks : 1) * sizeof(so_sb));
  for (uint32_t b = 0; b < Tm->n_blocks; b++) { bv[b].score = idf * Tm->blocks[b].max_part; bv[b].ord = b; }
  qsort(bv, Tm->n_blocks, sizeof(so_sb), sb_cmp); /* single.rs:378 */
  uint16_t* ids = (uint16_t*)malloc(65536 * sizeof(uint16_t));
  for (uint32_t bi = 0; bi < Tm->n_blocks; bi++) {
    if (!filtered && bi == top_k) break; /* single.rs:380-382 */
    const float bs = bv[bi].score;
    if (T->h.n == top_k && T->h.k && bs <= T->h.e[0].score) { /* single.rs:383-391 */
      if (!filtered) break;
      else if (rt == SO_RT_TOPK) continue;
    }
    /* single_docid: single.rs:41-50 */
    if ((rt == SO_RT_COUNT || (T->h.n == top_k && T->h.k && bs <= T->h.e[0].score)) && (!filtered || rt == SO_RT_TOPK)) continue;
    const so_blk* B = &Tm->blocks[bv[bi].ord];
    const uint32_t n = decode_block(B, ids);
    for (uint32_t p = 0; p < n; p++) {
      const uint32_t docid = (B->block_id << 16) | ids[p];
      if (gone && gone[docid]) continue; /* add_result.rs:661-663 delete_hashset, 665-727 not_query_list */
      if (rt == SO_RT_COUNT) { local++; continue; }
      if (rt == SO_RT_TOPKCOUNT) local++;
      if (T->h.n >= top_k && bs <= T->h.e[0].score) continue; /* add_result.rs:765-772, 829-836 */
      const float bm25 = so_bm25_term(idf, B->tf[p], s->comp[s->doclen[docid]]); /* add_result.rs:1090-1096 */
      ref_add_topk(T, docid, bm25);
    }
  }
  free(ids); free(bv);
  return filtered ? local : Tm->posting_count; /* single.rs:405-413 */
}

/* intersection_blockid through the hashset-aware heap: search_and's body with ref_add_topk.  (search_and above keeps the
 * plain heap for the direct intersection path.) */
static uint64_t ref_intersection(const so_shard* s, uint32_t nq, const uint32_t* qt, const float* idf, int rt, so_topk_ref* T,
                                 const uint8_t* gone) {
  /* the heap is shared: run search_and on a heap view whose add goes through the hashset.  search_and only calls
   * heap_add_topk / heap_full and reads e[0]; when the hashset is empty both adds are identical (min_heap.rs:1201), so the
   * plain call is exact then.  With entries present the candidates are collected and re-offered through ref_add_topk. */
  uint64_t total = 0;
  if (T->hs.n == 0) { search_and(s, nq, qt, idf, rt, &T->h, &total, gone); return total; }
  /* general case: replay search_and's loop with the hashset arm */
  uint32_t ptr[32] = {0};
  uint32_t nbm = 0, cap = 0;
  for (uint32_t t = 0; t < nq; t++) if (t == 0 || s->terms[qt[t]].n_blocks < cap) cap = s->terms[qt[t]].n_blocks;
  so_bm* bms = (so_bm*)malloc((cap ? cap : 1) * sizeof(so_bm));
  for (;;) {
    int done = 0; uint32_t mx = 0;
    for (uint32_t t = 0; t < nq; t++) {
      const so_term* Tm = &s->terms[qt[t]];
      if (ptr[t] >= Tm->n_blocks) { done = 1; break; }
      if (Tm->blocks[ptr[t]].block_id > mx) mx = Tm->blocks[ptr[t]].block_id;
    }
    if (done) break;
    int all = 1;
    for (uint32_t t = 0; t < nq; t++) {
      const so_term* Tm = &s->terms[qt[t]];
      while (ptr[t] < Tm->n_blocks && Tm->blocks[ptr[t]].block_id < mx) ptr[t]++;
      if (ptr[t] >= Tm->n_blocks) { done = 1; break; }
      if (Tm->blocks[ptr[t]].block_id != mx) all = 0;
    }
    if (done) break;
    if (!all) continue;
    so_bm* m = &bms[nbm++];
    m->block_id = mx; m->score = 0.0f;
    for (uint32_t t = 0; t < nq; t++) { m->ord[t] = ptr[t]; m->score += idf[t] * s->terms[qt[t]].blocks[ptr[t]].max_part; ptr[t]++; }
  }
  if (rt != SO_RT_COUNT) qsort(bms, nbm, sizeof(so_bm), bm_cmp);
  uint16_t* first = (uint16_t*)malloc(65536 * sizeof(uint16_t));
  so_heap* heap = &T->h;
  for (uint32_t bi = 0; bi < nbm; bi++) {
    so_bm* m = &bms[bi];
    if (rt == SO_RT_TOPK && heap_full(heap) && heap->k > 0 && m->score <= heap->e[0].score) break;
    uint32_t order[32];
    for (uint32_t t = 0; t < nq; t++) order[t] = t;
    for (uint32_t i = 1; i < nq; i++) {
      uint32_t x = order[i]; uint32_t j = i;
      for (; j > 0; j--) {
        const so_blk* a = &s->terms[qt[order[j - 1]]].blocks[m->ord[order[j - 1]]];
        const so_blk* b = &s->terms[qt[x]].blocks[m->ord[x]];
        int abm = a->ctype == SO_CT_BITMAP, bbm = b->ctype == SO_CT_BITMAP;
        int gt = (abm != bbm) ? (abm > bbm) : (a->count > b->count);
        if (!gt) break;
        order[j] = order[j - 1];
      }
      order[j] = x;
    }
    so_cur cur[32];
    for (uint32_t i = 0; i < nq; i++) {
      uint32_t t = order[i];
      cur[i].b = &s->terms[qt[t]].blocks[m->ord[t]];
      cur[i].idf = idf[t]; cur[i].pos = 0; cur[i].rle_rank = 0; cur[i].rank = 0;
    }
    uint32_t n0 = decode_block(cur[0].b, first);
    for (uint32_t p0 = 0; p0 < n0; p0++) {
      uint32_t d = first[p0];
      int ok = 1;
      for (uint32_t i = 1; i < nq && ok; i++) ok = cur_find(&cur[i], d);
      if (!ok) continue;
      cur[0].rank = p0;
      uint32_t docid = (m->block_id << 16) | d;
      if (gone && gone[docid]) continue;
      if (rt == SO_RT_COUNT) { total++; continue; }
      if (heap_full(heap) && heap->k > 0 && m->score <= heap->e[0].score) { if (rt == SO_RT_TOPKCOUNT) total++; continue; }
      float comp = s->comp[s->doclen[docid]];
      float bm25 = 0.0f;
      for (uint32_t i = 0; i < nq; i++) bm25 += so_bm25_term(cur[i].idf, cur[i].b->tf[cur[i].rank], comp);
      total++;
      ref_add_topk(T, docid, bm25);
    }
  }
  free(first); free(bms);
  return total;
}

static float ref_max_list_score(const so_shard* s, uint32_t term, float idf) { /* max_list_score = max block score, index.rs:3239 */
  float m = 0.0f;
  for (uint32_t b = 0; b < s->terms[term].n_blocks; b++) { float v = idf * s->terms[term].blocks[b].max_part; if (v > m) m = v; }
  return m;
}

/* union_docid_2, union.rs:1168-1305.  Returns result_count. */
static uint64_t ref_union2(const so_shard* s, const uint32_t* qt, const float* idf, int rt, uint32_t top_k, so_topk_ref* T,
                           const uint8_t* gone_not, const uint8_t* gone) {
  /* filtered (union.rs:1184): NOT terms / field filter -- NOT a delete set alone */
  const int filtered = gone_not != NULL;
  uint64_t count = 0;
  if (filtered) {
    so_topk_ref dummy; memset(&dummy, 0, sizeof dummy);
    count = ref_single(s, qt[0], idf[0], SO_RT_COUNT, top_k, &dummy, gone, 2) + ref_single(s, qt[1], idf[1], SO_RT_COUNT, top_k, &dummy, gone, 2);
  }
  const uint64_t inter = ref_intersection(s, 2, qt, idf, rt, T, gone);
  uint64_t local = filtered ? count : s->terms[qt[0]].posting_count + s->terms[qt[1]].posting_count;
  if (local > inter) local -= inter;
  if (rt == SO_RT_COUNT) return local;
  for (int i = 0; i < 2; i++)
    if (T->h.n < top_k || ref_max_list_score(s, qt[i], idf[i]) > T->h.e[0].score) {
      ref_snapshot(T);
      ref_single(s, qt[i], idf[i], SO_RT_TOPK, top_k, T, gone, 2);
    }
  return local;
}

typedef struct { uint32_t n, query_index; float max_score; uint32_t term[10]; float idf[10]; } so_qobj;
/* union_docid_3, union.rs:1308-1479: the queue of sub-queries, best upper bound first; recursion_count < 200 */
static void ref_union3(const so_shard* s, uint32_t nq, const uint32_t* qt, const float* idf, uint32_t top_k, so_topk_ref* T,
                       const uint8_t* gone_not, const uint8_t* gone) {
  uint32_t qcap = 64, qn = 1;
  so_qobj* queue = (so_qobj*)malloc(qcap * sizeof(so_qobj));
  queue[0].n = nq; queue[0].query_index = 0; queue[0].max_score = 3.4028235e38f;
  memcpy(queue[0].term, qt, nq * sizeof(uint32_t)); memcpy(queue[0].idf, idf, nq * sizeof(float));
  for (uint32_t rec = 0; rec <= 200 && qn; rec++) {
    so_qobj q = queue[0];
    memmove(queue, queue + 1, (--qn) * sizeof(so_qobj)); /* query_queue.remove(0) */
    if (q.n >= 3) {
      ref_intersection(s, q.n, q.term, q.idf, SO_RT_TOPK, T, gone);
      ref_snapshot(T);
      for (uint32_t i = q.query_index; i < q.n; i++) {
        const uint32_t ii = q.n - 1 - i;
        so_qobj l; l.n = 0; l.query_index = i; l.max_score = 0.0f;
        for (uint32_t j = 0; j < q.n; j++) if (j != ii) { l.term[l.n] = q.term[j]; l.idf[l.n] = q.idf[j]; l.n++; }
        for (uint32_t j = 0; j < l.n; j++) l.max_score += ref_max_list_score(s, l.term[j], l.idf[j]);
        if (T->h.n < top_k || l.max_score > T->h.e[0].score) {
          if (qn == qcap) { qcap *= 2; queue = (so_qobj*)realloc(queue, qcap * sizeof(so_qobj)); }
          uint32_t pos = qn;
          if (qn && l.max_score > queue[qn - 1].max_score) { /* binary_search_by descending -> insertion point (any among equals) */
            uint32_t lo = 0, hi = qn;
            while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (queue[mid].max_score > l.max_score) lo = mid + 1; else hi = mid; }
            pos = lo;
          }
          memmove(queue + pos + 1, queue + pos, (qn - pos) * sizeof(so_qobj));
          queue[pos] = l; qn++;
        }
      }
    } else {
      ref_union2(s, q.term, q.idf, SO_RT_TOPK, top_k, T, gone_not, gone);
    }
    if (!(qn && (T->h.n < top_k || queue[0].max_score > T->h.e[0].score))) break;
    ref_snapshot(T);
  }
  free(queue);
}

/* search_lexical_shard's dispatch as the reference structures it (one indexed field, no field / facet filters, no sort).
 * Same signature as so_search_lex_not.  Known reference quirk kept: a 2-term union's count under a delete set WITHOUT NOT
 * terms is posting_count sums minus the intersection count (union.rs:1240-1249) and so still counts deleted docs. */
uint32_t so_search_lex_ref(const so_shard* s, uint32_t nq, const uint32_t* qt, uint32_t n_not, const uint32_t* not_terms,
                           int op, uint32_t k, int rt, uint32_t* od, float* os, uint64_t* total) {
  uint64_t tot = 0;
  if (nq == 0 || nq > 32) { if (total) *total = 0; return 0; }
  float idf[32];
  for (uint32_t t = 0; t < nq; t++) {
    if (qt[t] >= s->n_terms) { if (total) *total = 0; return 0; }
    idf[t] = so_idf(s->n_docs, s->terms[qt[t]].posting_count);
  }
  uint32_t kk = k; if ((uint64_t)kk > s->n_docs) kk = (uint32_t)s->n_docs;
  if (rt == SO_RT_COUNT) kk = 0;
  so_topk_ref T;
  T.h.n = 0; T.h.k = kk; T.h.e = (so_res*)malloc((kk ? kk : 1) * sizeof(so_res));
  hset_init(&T.hs, 256);
  uint8_t* gone = exclusion_map(s, n_not, not_terms);
  const uint8_t* gone_not = n_not ? gone : NULL;
  if (nq == 1) tot = ref_single(s, qt[0], idf[0], rt, kk, &T, gone, 1);
  else if (op == SO_OP_AND) search_and(s, nq, qt, idf, rt, &T.h, &tot, gone);
  else if (rt == SO_RT_COUNT && nq != 2) search_or(s, nq, qt, idf, rt, &T.h, &tot, gone);
  else if (nq == 2) tot = ref_union2(s, qt, idf, rt, kk, &T, gone_not, gone);
  else if (nq <= 10) {
    ref_union3(s, nq, qt, idf, kk, &T, gone_not, gone);
    if (rt == SO_RT_TOPKCOUNT) { /* union.rs:1455-1477: union_blockid with ResultType::Count */
      so_heap none; none.n = 0; none.k = 0; none.e = NULL;
      search_or(s, nq, qt, idf, SO_RT_COUNT, &none, &tot, gone);
    }
  } else search_or(s, nq, qt, idf, rt, &T.h, &tot, gone);
  uint32_t n = heap_drain(&T.h, od, os);
  free(gone); free(T.h.e); hset_free(&T.hs);
  if (total) *total = tot;
  return n;
}

/* ================================================================== CPU baseline harness (bench.py cpu_baseline leg)
 * The reference's execution structure for one query over an index of S document-partitioned shards: one task per shard
 * (search.rs:1637-1743), each single-threaded inside its shard, then gather + sort + truncate (search.rs:1875-1940,
 * 2098-2119), global id = local * S + shard (search.rs:1671).
 *   mode 0 (throughput): `threads` workers, each answers whole queries (its S shard tasks one after the other, then the
 *                        merge) -- every core busy with independent queries;
 *   mode 1 (latency):    one query at a time, its S shard tasks on S worker threads, merge on the caller -> per-query
 *                        wall time (p50 / p99 come from out_lat_us).
 * Queries: q_terms [nq][nt] term ids valid in every shard.  Runs for about `seconds`, cycling through the queries. */
#include <pthread.h>
#include <stdatomic.h>
#include <time.h>
#include <sched.h>
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
typedef struct { uint64_t doc; float score; } so_gres;
static int gres_cmp(const void* a, const void* b) {
  const so_gres* x = (const so_gres*)a; const so_gres* y = (const so_gres*)b;
  if (x->score > y->score) return -1;
  if (x->score < y->score) return 1;
  return 0;
}
typedef struct {
  so_shard* const* shards; uint32_t S; const uint32_t* q; uint32_t nq, nt; int op; uint32_t k; int rt; double t_end;
  atomic_uint_fast64_t next, done;
  /* latency mode */
  atomic_uint_fast64_t gen; atomic_uint_fast32_t pending; atomic_int stop; uint32_t cur_q;
  uint32_t* r_doc; float* r_score; uint32_t* r_n; /* [S][k] per-shard results */
  uint64_t checksum;
} so_bctx;
typedef struct { so_bctx* c; uint32_t id; } so_barg;
static void shard_task(so_bctx* c, uint32_t sh, uint32_t qi, uint32_t* od, float* os, uint32_t* n) {
  uint64_t tot;
  *n = so_search_lex_ref(c->shards[sh], c->nt, c->q + (size_t)qi * c->nt, 0, NULL, c->op, c->k, c->rt, od, os, &tot);
}
static uint64_t merge_shards(so_bctx* c, const uint32_t* r_doc, const float* r_score, const uint32_t* r_n, so_gres* tmp) {
  uint32_t m = 0;
  for (uint32_t sh = 0; sh < c->S; sh++)
    for (uint32_t i = 0; i < r_n[sh]; i++) { tmp[m].doc = (uint64_t)r_doc[(size_t)sh * c->k + i] * c->S + sh; tmp[m].score = r_score[(size_t)sh * c->k + i]; m++; }
  qsort(tmp, m, sizeof(so_gres), gres_cmp);
  return m ? tmp[0].doc : 0;
}
static void* thr_throughput(void* a_) {
  so_barg* a = (so_barg*)a_; so_bctx* c = a->c;
  uint32_t* od = (uint32_t*)malloc((size_t)c->S * c->k * sizeof(uint32_t));
  float* os = (float*)malloc((size_t)c->S * c->k * sizeof(float));
  uint32_t* n = (uint32_t*)malloc(c->S * sizeof(uint32_t));
  so_gres* tmp = (so_gres*)malloc((size_t)c->S * c->k * sizeof(so_gres));
  uint64_t chk = 0;
  while (now_s() < c->t_end) {
    const uint64_t i = atomic_fetch_add(&c->next, 1);
    const uint32_t qi = (uint32_t)(i % c->nq);
    for (uint32_t sh = 0; sh < c->S; sh++) shard_task(c, sh, qi, od + (size_t)sh * c->k, os + (size_t)sh * c->k, n + sh);
    chk += merge_shards(c, od, os, n, tmp);
    atomic_fetch_add(&c->done, 1);
  }
  __atomic_fetch_add(&c->checksum, chk, __ATOMIC_RELAXED);
  free(od); free(os); free(n); free(tmp);
  return NULL;
}
static void* thr_latency(void* a_) {
  so_barg* a = (so_barg*)a_; so_bctx* c = a->c;
  uint64_t seen = 0;
  for (;;) {
    uint64_t g;
    uint32_t spins = 0;
    while ((g = atomic_load_explicit(&c->gen, memory_order_acquire)) == seen) {
      if (atomic_load_explicit(&c->stop, memory_order_relaxed)) return NULL;
      if (++spins > 2000) { sched_yield(); spins = 0; }
    }
    seen = g;
    shard_task(c, a->id, c->cur_q, c->r_doc + (size_t)a->id * c->k, c->r_score + (size_t)a->id * c->k, c->r_n + a->id);
    atomic_fetch_sub_explicit(&c->pending, 1, memory_order_release);
  }
}
/* returns queries per second; *out_queries = queries answered; latency mode fills out_lat_us[0 .. *out_nlat) */
double so_bench_lex(so_shard* const* shards, uint32_t S, const uint32_t* q_terms, uint32_t nq, uint32_t nt, int op, uint32_t k,
                    int rt, int mode, uint32_t threads, double seconds, uint64_t* out_queries, double* out_lat_us,
                    uint32_t lat_cap, uint32_t* out_nlat) {
  so_bctx c; memset(&c, 0, sizeof c);
  c.shards = shards; c.S = S; c.q = q_terms; c.nq = nq; c.nt = nt; c.op = op; c.k = k; c.rt = rt;
  atomic_init(&c.next, 0); atomic_init(&c.done, 0); atomic_init(&c.gen, 0); atomic_init(&c.pending, 0); atomic_init(&c.stop, 0);
  const uint32_t nthr = mode == 0 ? threads : S;
  pthread_t* th = (pthread_t*)malloc(nthr * sizeof(pthread_t));
  so_barg* args = (so_barg*)malloc(nthr * sizeof(so_barg));
  double t0 = now_s(), el;
  uint32_t nlat = 0;
  if (mode == 0) {
    c.t_end = t0 + seconds;
    for (uint32_t i = 0; i < nthr; i++) { args[i].c = &c; args[i].id = i; pthread_create(&th[i], NULL, thr_throughput, &args[i]); }
    for (uint32_t i = 0; i < nthr; i++) pthread_join(th[i], NULL);
    el = now_s() - t0;
  } else {
    c.r_doc = (uint32_t*)malloc((size_t)S * k * sizeof(uint32_t)); c.r_score = (float*)malloc((size_t)S * k * sizeof(float));
    c.r_n = (uint32_t*)calloc(S, sizeof(uint32_t));
    so_gres* tmp = (so_gres*)malloc((size_t)S * k * sizeof(so_gres));
    for (uint32_t i = 0; i < nthr; i++) { args[i].c = &c; args[i].id = i; pthread_create(&th[i], NULL, thr_latency, &args[i]); }
    uint64_t qn = 0;
    t0 = now_s();
    while (now_s() - t0 < seconds) {
      const double a = now_s();
      c.cur_q = (uint32_t)(qn % nq);
      atomic_store_explicit(&c.pending, S, memory_order_relaxed);
      atomic_fetch_add_explicit(&c.gen, 1, memory_order_release);
      uint32_t spins = 0;
      while (atomic_load_explicit(&c.pending, memory_order_acquire)) if (++spins > 2000) { sched_yield(); spins = 0; }
      c.checksum += merge_shards(&c, c.r_doc, c.r_score, c.r_n, tmp);
      const double b = now_s();
      if (out_lat_us && nlat < lat_cap) out_lat_us[nlat++] = (b - a) * 1e6;
      qn++;
    }
    el = now_s() - t0;
    atomic_store(&c.stop, 1);
    for (uint32_t i = 0; i < nthr; i++) pthread_join(th[i], NULL);
    atomic_store(&c.done, qn);
    free(c.r_doc); free(c.r_score); free(c.r_n); free(tmp);
  }
  free(th); free(args);
  const uint64_t done = atomic_load(&c.done);
  if (out_queries) *out_queries = done;
  if (out_nlat) *out_nlat = nlat;
  return el > 0 ? (double)done / el : 0.0;
}


/* ================================================================== phrase queries (QueryType::Phrase)
 * The phrase check of add_result_multiterm_singlefield (add_result.rs:3586-3684): the positions of every phrase entry
 * (non_unique_query_list: one entry per word of the phrase, term_index_nonunique = its place in the phrase; repeated words
 * share a unique term's positions) are walked by a merge on the phrase start pos - term_index_nonunique, the entry with the
 * fewest positions first (sort at 3616-3617).  Positions are decoded from their delta form first + (delta + 1) ...
 * (get_next_position_singlefield, add_result.rs:38-59; "pos1 += next + 1" at 3640, 3678).  A doc matches when some start
 * carries every word at its place.  Two restatements: the reference's merge loop (phrase_match_ref) and the definition
 * (phrase_match_def); tests assert they agree. */
/* counts (optional): positions per posting where that is not the tf -- the lists an N-GRAM key is held as (one per component term,
 * same docs, the component's tf, add_result.rs:2074-2089): the key's own positions_count / positions belong to the key, here kept
 * behind the FIRST component's postings (count 0 for the others). */
void so_shard_set_positions_counts(so_shard* s, const uint16_t* positions, uint64_t n_positions, const uint16_t* counts) {
  free(s->pos_off); free(s->pos);
  const uint64_t np = s->off[s->n_terms];
  s->pos_off = (uint64_t*)malloc((np + 1) * sizeof(uint64_t));
  uint64_t a = 0;
  for (uint64_t i = 0; i < np; i++) { s->pos_off[i] = a; a += counts ? counts[i] : s->tfs[i]; }
  s->pos_off[np] = a;
  if (a != n_positions) { free(s->pos_off); s->pos_off = NULL; s->pos = NULL; return; }
  s->pos = (uint16_t*)malloc((a ? a : 1) * sizeof(uint16_t));
  memcpy(s->pos, positions, a * sizeof(uint16_t));
}
void so_shard_set_positions(so_shard* s, const uint16_t* positions, uint64_t n_positions) {
  so_shard_set_positions_counts(s, positions, n_positions, NULL);
}
/* definition: exists start with entry i at start + place[i] for every i (place[i] = term_index_nonunique: i for a phrase of single
 * terms; an n-gram key is one entry spanning 2 / 3 places, search.rs:3305-3328) */
static int phrase_match_def(uint32_t n_seq, const uint16_t* const* pos, const uint32_t* cnt, const uint32_t* place) {
  for (uint32_t j = 0; j < cnt[0]; j++) {
    const uint32_t start = pos[0][j];
    int ok = 1;
    for (uint32_t i = 1; i < n_seq && ok; i++) {
      ok = 0;
      for (uint32_t x = 0; x < cnt[i]; x++) if ((uint32_t)pos[i][x] + place[0] == start + place[i]) { ok = 1; break; }
    }
    if (ok) return 1;
  }
  return 0;
}
/* the reference's loop, add_result.rs:3596-3684 (phrasematch_count >= 1 ends it) */
typedef struct { uint32_t idx, count, p_pos, pos; const uint16_t* list; } so_nu;
static int nu_cmp(const void* a, const void* b) { /* sort_unstable_by positions_count; ties: by idx (any order is legal) */
  const so_nu* x = (const so_nu*)a; const so_nu* y = (const so_nu*)b;
  if (x->count != y->count) return x->count < y->count ? -1 : 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}
static int phrase_match_ref(uint32_t n_seq, const uint16_t* const* pos, const uint32_t* cnt, const uint32_t* place) {
  so_nu nu[32];
  if (n_seq < 2) return cnt[0] > 0;
  for (uint32_t i = 0; i < n_seq; i++) {
    if (cnt[i] == 0) return 0;
    nu[i].idx = place[i]; nu[i].count = cnt[i]; nu[i].p_pos = 0; nu[i].list = pos[i]; nu[i].pos = pos[i][0];
  }
  qsort(nu, n_seq, sizeof(so_nu), nu_cmp);
  uint32_t t2 = 1;
  uint32_t pos1 = nu[0].pos, pos2 = nu[1].pos;
  for (;;) {
    const uint32_t l = pos1 + nu[t2].idx, r = pos2 + nu[0].idx;
    if (l < r) {
      if (t2 > 1) { t2 = 1; pos2 = nu[t2].pos; }
      if (++nu[0].p_pos == nu[0].count) return 0;
      pos1 = nu[0].list[nu[0].p_pos]; /* pos1 += delta + 1: the next absolute position */
    } else if (l > r) {
      if (++nu[t2].p_pos == nu[t2].count) return 0;
      pos2 = nu[t2].list[nu[t2].p_pos];
      nu[t2].pos = pos2;
    } else {
      if (t2 + 1 < n_seq) { t2++; pos2 = nu[t2].pos; continue; }
      return 1;
    }
  }
}
/* place: the entries' places in the phrase (term_index_nonunique), NULL = 0, 1, 2, ... */
int so_phrase_match_places(uint32_t n_seq, const uint16_t* const* pos, const uint32_t* cnt, const uint32_t* place, int reference_loop) {
  uint32_t ident[32];
  if (n_seq > 32) return 0;
  if (!place) { for (uint32_t i = 0; i < n_seq; i++) ident[i] = i; place = ident; }
  return reference_loop ? phrase_match_ref(n_seq, pos, cnt, place) : phrase_match_def(n_seq, pos, cnt, place);
}
int so_phrase_match(uint32_t n_seq, const uint16_t* const* pos, const uint32_t* cnt, int reference_loop) {
  return so_phrase_match_places(n_seq, pos, cnt, NULL, reference_loop);
}

/* Phrase search over one indexed field: the docs containing every unique term (intersection) whose positions carry the
 * phrase; scored like the intersection (get_bm25f_multiterm_singlefield over the unique terms, add_result.rs:3573, 3692),
 * counted only when the phrase matches (3686-3690).  seq[i] = index into q_terms of the i-th word.  Exact top-k by (score
 * desc, doc asc); *total = matches. */
uint32_t so_search_phrase(const so_shard* s, uint32_t nq, const uint32_t* qt, uint32_t n_seq, const uint8_t* seq, uint32_t k,
                          int reference_loop, uint32_t* od, float* os, uint64_t* total) {
  return so_search_phrase_items(s, nq, qt, NULL, n_seq, seq, NULL, k, reference_loop, od, os, total);
}
/* ... with N-GRAM keys among the phrase's entries (the reference's default index, index.rs:1422-1424).  The unique terms qt hold,
 * for an n-gram key, its 2 / 3 component lists (scored with idf_in[t] = idf_ngram_i, search.rs:3231-3262; NULL = every idf from
 * the list's own posting count); entry i of the phrase = unique term seq[i] (an n-gram key: its FIRST component, which carries the
 * key's positions, so_shard_set_positions_counts) at place[i] = entries before it + the extra places of the n-gram keys before it
 * (term_index_nonunique, search.rs:3305-3328). */
uint32_t so_search_phrase_items(const so_shard* s, uint32_t nq, const uint32_t* qt, const float* idf_in, uint32_t n_seq, const uint8_t* seq,
                                const uint8_t* place_in, uint32_t k, int reference_loop, uint32_t* od, float* os, uint64_t* total) {
  if (!s->pos || nq == 0 || nq > 32 || n_seq == 0 || n_seq > 32) { if (total) *total = 0; return 0; }
  float idf[32];
  uint32_t place[32];
  for (uint32_t i = 0; i < n_seq; i++) place[i] = place_in ? place_in[i] : i;
  for (uint32_t t = 0; t < nq; t++) idf[t] = idf_in ? idf_in[t] : so_idf(s->n_docs, s->terms[qt[t]].posting_count);
  uint64_t cur[32];
  for (uint32_t t = 0; t < nq; t++) cur[t] = s->off[qt[t]];
  so_sd* v = NULL; uint64_t nv = 0, cap = 0;
  for (uint64_t i0 = s->off[qt[0]]; i0 < s->off[qt[0] + 1]; i0++) {
    const uint32_t d = s->docs[i0];
    int all = 1;
    uint64_t at[32]; at[0] = i0;
    for (uint32_t t = 1; t < nq && all; t++) {
      while (cur[t] < s->off[qt[t] + 1] && s->docs[cur[t]] < d) cur[t]++;
      all = cur[t] < s->off[qt[t] + 1] && s->docs[cur[t]] == d;
      at[t] = cur[t];
    }
    if (!all) continue;
    if (s->deleted && s->deleted[d]) continue;
    const uint16_t* pl[32]; uint32_t pc[32];
    for (uint32_t i = 0; i < n_seq; i++) { pl[i] = s->pos + s->pos_off[at[seq[i]]]; pc[i] = (uint32_t)(s->pos_off[at[seq[i]] + 1] - s->pos_off[at[seq[i]]]); }
    if (n_seq >= 2 && !so_phrase_match_places(n_seq, pl, pc, place, reference_loop)) continue;  /* one entry: a term query (search.rs:3544) */
    float sc = 0.0f;
    const float comp = s->comp[s->doclen[d]];
    for (uint32_t t = 0; t < nq; t++) sc += so_bm25_term(idf[t], s->tfs[at[t]], comp);
    if (nv == cap) { cap = cap ? cap * 2 : 1024; v = (so_sd*)realloc(v, cap * sizeof(so_sd)); }
    v[nv].score = sc; v[nv].doc = d; nv++;
  }
  if (nv) qsort(v, nv, sizeof(so_sd), sd_cmp);
  const uint32_t n = (uint32_t)(nv < k ? nv : k);
  for (uint32_t i = 0; i < n; i++) { od[i] = v[i].doc; os[i] = v[i].score; }
  if (total) *total = nv;
  free(v);
  return n;
}

/* Phrase search over SEVERAL indexed fields (add_result_multiterm_multifield, add_result.rs:2964-3414): the docs containing every
 * unique term in some field (intersection of the keys' posting lists); the phrase check runs FIELD BY FIELD, ascending
 * ('main loop, 3259-3386): a field is looked at only if every word of the phrase has positions in it (3260-3277: a word whose
 * next field is larger skips the field, a word without further fields ends the search) and, under a field filter, only if the
 * filter lists it (3285-3287); inside the field it is the single-field merge over that field's positions (3289-3385;
 * positions restart in every field: get_next_position_multifield is absolute at the first position of a field, 3279-3283).
 * The first match ends it (phrasematch_count >= 1 -> break 'main).  A matching doc is counted (3394-3396) and scored with
 * get_bm25f_multiterm_multifield over ALL fields of the unique terms (3140 / 3402; 1226-1262: terms in query order, a term's
 * fields ascending, boost * idf * (tf (K+1) / (tf + comp[len byte of (doc, field)]) + SIGMA)) -- the filter restricts where the
 * phrase may stand, not what is summed (the per-term gate 3124-3136 is implied by a match inside a listed field).
 * Entries of a term sorted by (doc, field); positions: for every entry in CSR order its tf positions inside the field,
 * ascending; seq[i] = index into qt of the i-th word; field_mask 0 = no filter.  Exact top-k by (score desc, doc asc). */
uint32_t so_search_fields_phrase(uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen /*[n_fields][n_docs]*/, const float* boost,
                                 const uint64_t* off, const uint32_t* docs, const uint8_t* fields, const uint16_t* tfs,
                                 const uint16_t* positions, uint32_t nq, const uint32_t* qt, uint32_t n_seq, const uint8_t* seq,
                                 uint32_t k, const uint64_t* deleted, uint64_t n_deleted, uint32_t field_mask, int reference_loop,
                                 uint32_t* od, float* os, uint64_t* total) {
  return so_search_fields_phrase_items(n_docs, n_fields, doclen, boost, off, docs, fields, tfs, NULL, positions, nq, qt, NULL, n_seq, seq, NULL, k,
                                       deleted, n_deleted, field_mask, reference_loop, od, os, total);
}
/* ... with N-GRAM keys among the phrase's entries (several indexed fields: add_result.rs:1524-1600 reads the components' field vectors,
 * then the key's own vector and positions).  The lists qt hold, for an n-gram key, its 2 / 3 component lists -- entries (doc, field,
 * tf of the component in the field) --, counts[e] = positions behind entry e: the key's own count in that field with its FIRST
 * component, 0 with the others and where the key does not stand in the field (NULL = tfs: SingleTerm lists only); idf_in per list
 * (idf_ngram_i; NULL = from the list's docs); place[i] = term_index_nonunique of entry i (NULL = 0, 1, ...). */
uint32_t so_search_fields_phrase_items(uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen, const float* boost, const uint64_t* off,
                                       const uint32_t* docs, const uint8_t* fields, const uint16_t* tfs, const uint16_t* counts,
                                       const uint16_t* positions, uint32_t nq, const uint32_t* qt, const float* idf_in, uint32_t n_seq,
                                       const uint8_t* seq, const uint8_t* place_in, uint32_t k, const uint64_t* deleted, uint64_t n_deleted,
                                       uint32_t field_mask, int reference_loop, uint32_t* od, float* os, uint64_t* total) {
  if (total) *total = 0;
  if (!counts) counts = tfs;
  uint32_t place[32];
  for (uint32_t i = 0; i < n_seq && i < 32; i++) place[i] = place_in ? place_in[i] : i;
  if (nq == 0 || nq > 32 || n_seq < 2 || n_seq > 32 || n_fields == 0 || n_fields > 16) return 0;
  uint64_t psum = 0;
  for (uint64_t i = 0; i < n_docs * n_fields; i++) psum += so_byte4_to_int(doclen[i]);
  float comp[256];
  so_bm25_component_cache(so_avgdl(psum, n_docs), comp);
  /* first position of every entry */
  uint64_t last = 0;
  for (uint32_t t = 0; t < nq; t++) if (off[qt[t] + 1] > last) last = off[qt[t] + 1];
  uint64_t* pbeg = (uint64_t*)malloc((last + 1) * sizeof(uint64_t));
  { uint64_t a = 0; for (uint64_t i = 0; i < last; i++) { pbeg[i] = a; a += counts[i]; } pbeg[last] = a; }
  uint8_t* dead = (uint8_t*)calloc(n_docs ? n_docs : 1, 1);
  for (uint64_t i = 0; i < n_deleted; i++) if (deleted[i] < n_docs) dead[deleted[i]] = 1;
  float idf[32];
  for (uint32_t t = 0; t < nq; t++) {
    uint64_t df = 0;
    for (uint64_t i = off[qt[t]]; i < off[qt[t] + 1]; i++) if (i == off[qt[t]] || docs[i] != docs[i - 1]) df++;
    idf[t] = idf_in ? idf_in[t] : so_idf(n_docs, df);
  }
  uint64_t cur[32];
  for (uint32_t t = 0; t < nq; t++) cur[t] = off[qt[t]];
  so_sd* v = NULL; uint64_t nv = 0, cap = 0;
  for (uint64_t i0 = off[qt[0]]; i0 < off[qt[0] + 1]; i0++) {
    if (i0 > off[qt[0]] && docs[i0] == docs[i0 - 1]) continue;  /* first entry of (term 0, doc) */
    const uint32_t d = docs[i0];
    int all = 1;
    uint64_t at[32], end[32];  /* entries of (term, doc): at[t] .. end[t] */
    at[0] = i0;
    for (uint32_t t = 1; t < nq && all; t++) {
      while (cur[t] < off[qt[t] + 1] && docs[cur[t]] < d) cur[t]++;
      all = cur[t] < off[qt[t] + 1] && docs[cur[t]] == d;
      at[t] = cur[t];
    }
    if (!all || dead[d]) continue;
    for (uint32_t t = 0; t < nq; t++) { end[t] = at[t]; while (end[t] < off[qt[t] + 1] && docs[end[t]] == d) end[t]++; }
    int match = 0;
    for (uint32_t f = 0; f < n_fields && !match; f++) {
      const uint16_t* pl[32]; uint32_t pc[32];
      int have = 1;
      for (uint32_t i = 0; i < n_seq && have; i++) {
        const uint32_t t = seq[i];
        have = 0;
        for (uint64_t e = at[t]; e < end[t]; e++)
          if (fields[e] == f && counts[e]) { pl[i] = positions + pbeg[e]; pc[i] = counts[e]; have = 1; break; }
      }
      if (!have) continue;                                        /* some word has no position in this field */
      if (field_mask && !((field_mask >> f) & 1u)) continue;      /* add_result.rs:3285-3287 */
      match = so_phrase_match_places(n_seq, pl, pc, place, reference_loop);
    }
    if (!match) continue;
    float sc = 0.0f;
    for (uint32_t t = 0; t < nq; t++)
      for (uint64_t e = at[t]; e < end[t]; e++) {
        const float w = boost ? boost[fields[e]] : 1.0f;
        sc += w * idf[t] * ((float)tfs[e] * (SO_K + 1.0f) / ((float)tfs[e] + comp[doclen[(uint64_t)fields[e] * n_docs + d]]) + SO_SIGMA);
      }
    if (nv == cap) { cap = cap ? cap * 2 : 1024; v = (so_sd*)realloc(v, cap * sizeof(so_sd)); }
    v[nv].score = sc; v[nv].doc = d; nv++;
  }
  if (nv) qsort(v, nv, sizeof(so_sd), sd_cmp);
  const uint32_t n = (uint32_t)(nv < k ? nv : k);
  for (uint32_t i = 0; i < n; i++) { od[i] = v[i].doc; os[i] = v[i].score; }
  if (total) *total = nv;
  free(v); free(dead); free(pbeg);
  return n;
}

/* ================================================================== CPU baseline harness, vector path (bench.py)
 * search_vector_shard's AnnMode::All scan (read_record -> dot_f32_avx2 -> TopK::push, vector.rs:1397-1466) in the reference's
 * execution structure: the records partitioned over S shards, one task per shard and query, gather + sort.
 *   mode 0 (throughput): `threads` workers, each answers whole queries over ALL rows;
 *   mode 1 (latency):    one query at a time, `threads` workers each scanning its slice of the rows, merge on the caller.
 * rows [n_rows][dim] f32, queries [nq][dim].  Returns queries/s. */
typedef struct {
  const float* rows; uint64_t n_rows; uint32_t dim; const float* q; uint32_t nq, k; double t_end;
  atomic_uint_fast64_t next, done, gen; atomic_uint_fast32_t pending; atomic_int stop;
  uint32_t cur_q, S; uint32_t* r_doc; float* r_score; uint32_t* r_n; uint64_t checksum;
} so_vctx;
typedef struct { so_vctx* c; uint32_t id; } so_varg;
static void* vthr_throughput(void* a_) {
  so_varg* a = (so_varg*)a_; so_vctx* c = a->c;
  uint32_t* od = (uint32_t*)malloc(c->k * sizeof(uint32_t));
  float* os = (float*)malloc(c->k * sizeof(float));
  uint64_t chk = 0;
  while (now_s() < c->t_end) {
    const uint64_t i = atomic_fetch_add(&c->next, 1);
    so_f32_ctx x = {c->rows, c->q + (size_t)(i % c->nq) * c->dim, c->dim, 1, 0};
    uint64_t tot, obs;
    uint32_t n = topk_scan(c->n_rows, NULL, c->k, -3.4028235e38f, NULL, 0, score_f32, &x, od, os, &tot, &obs);
    chk += n ? od[0] : 0;
    atomic_fetch_add(&c->done, 1);
  }
  __atomic_fetch_add(&c->checksum, chk, __ATOMIC_RELAXED);
  free(od); free(os);
  return NULL;
}
static void* vthr_latency(void* a_) {
  so_varg* a = (so_varg*)a_; so_vctx* c = a->c;
  const uint64_t r0 = c->n_rows * a->id / c->S, r1 = c->n_rows * (a->id + 1) / c->S;
  uint64_t seen = 0;
  for (;;) {
    uint64_t g; uint32_t spins = 0;
    while ((g = atomic_load_explicit(&c->gen, memory_order_acquire)) == seen) {
      if (atomic_load_explicit(&c->stop, memory_order_relaxed)) return NULL;
      if (++spins > 2000) { sched_yield(); spins = 0; }
    }
    seen = g;
    so_f32_ctx x = {c->rows + r0 * c->dim, c->q + (size_t)c->cur_q * c->dim, c->dim, 1, 0};
    uint64_t tot, obs;
    c->r_n[a->id] = topk_scan(r1 - r0, NULL, c->k, -3.4028235e38f, NULL, 0, score_f32, &x, c->r_doc + (size_t)a->id * c->k,
                              c->r_score + (size_t)a->id * c->k, &tot, &obs);
    atomic_fetch_sub_explicit(&c->pending, 1, memory_order_release);
  }
}
double so_bench_vec(const float* rows, uint64_t n_rows, uint32_t dim, const float* queries, uint32_t nq, uint32_t k, int mode,
                    uint32_t threads, double seconds, uint64_t* out_queries, double* out_lat_us, uint32_t lat_cap, uint32_t* out_nlat) {
  so_vctx c; memset(&c, 0, sizeof c);
  c.rows = rows; c.n_rows = n_rows; c.dim = dim; c.q = queries; c.nq = nq; c.k = k; c.S = threads;
  atomic_init(&c.next, 0); atomic_init(&c.done, 0); atomic_init(&c.gen, 0); atomic_init(&c.pending, 0); atomic_init(&c.stop, 0);
  pthread_t* th = (pthread_t*)malloc(threads * sizeof(pthread_t));
  so_varg* args = (so_varg*)malloc(threads * sizeof(so_varg));
  double t0 = now_s(), el;
  uint32_t nlat = 0;
  if (mode == 0) {
    c.t_end = t0 + seconds;
    for (uint32_t i = 0; i < threads; i++) { args[i].c = &c; args[i].id = i; pthread_create(&th[i], NULL, vthr_throughput, &args[i]); }
    for (uint32_t i = 0; i < threads; i++) pthread_join(th[i], NULL);
    el = now_s() - t0;
  } else {
    c.r_doc = (uint32_t*)malloc((size_t)threads * k * sizeof(uint32_t)); c.r_score = (float*)malloc((size_t)threads * k * sizeof(float));
    c.r_n = (uint32_t*)calloc(threads, sizeof(uint32_t));
    so_gres* tmp = (so_gres*)malloc((size_t)threads * k * sizeof(so_gres));
    for (uint32_t i = 0; i < threads; i++) { args[i].c = &c; args[i].id = i; pthread_create(&th[i], NULL, vthr_latency, &args[i]); }
    uint64_t qn = 0;
    t0 = now_s();
    while (now_s() - t0 < seconds) {
      const double a = now_s();
      c.cur_q = (uint32_t)(qn % nq);
      atomic_store_explicit(&c.pending, threads, memory_order_relaxed);
      atomic_fetch_add_explicit(&c.gen, 1, memory_order_release);
      uint32_t spins = 0;
      while (atomic_load_explicit(&c.pending, memory_order_acquire)) if (++spins > 2000) { sched_yield(); spins = 0; }
      uint32_t m = 0;
      for (uint32_t sh = 0; sh < threads; sh++)
        for (uint32_t i = 0; i < c.r_n[sh]; i++) { tmp[m].doc = c.r_doc[(size_t)sh * k + i]; tmp[m].score = c.r_score[(size_t)sh * k + i]; m++; }
      qsort(tmp, m, sizeof(so_gres), gres_cmp);
      c.checksum += m ? tmp[0].doc : 0;
      const double b = now_s();
      if (out_lat_us && nlat < lat_cap) out_lat_us[nlat++] = (b - a) * 1e6;
      qn++;
    }
    el = now_s() - t0;
    atomic_store(&c.stop, 1);
    for (uint32_t i = 0; i < threads; i++) pthread_join(th[i], NULL);
    atomic_store(&c.done, qn);
    free(c.r_doc); free(c.r_score); free(c.r_n); free(tmp);
  }
  free(th); free(args);
  const uint64_t done = atomic_load(&c.done);
  if (out_queries) *out_queries = done;
  if (out_nlat) *out_nlat = nlat;
  return el > 0 ? (double)done / el : 0.0;
}

/* ================================================================== TurboQuant (Quantization::TurboQuantI8)
 * TurboQuant::quantize_f32_i8 (vector_similarity.rs:1927-1956) and its AVX2 form (1958-1983): pad to the next power of two,
 * multiply by the +-1 seed mask, Fast Walsh-Hadamard transform (1861-1879 / 1883-1925: butterflies, then / sqrt(n)),
 * scale = max(sqrt(sum x^2) / sqrt(dim) / 32, 1e-8) (2011-2039), q = round(x / scale) clamped to +-127, norm = sum q^2 * scale^2.
 * avx2 != 0: the sum of squares in eight lanes (mul, then add) folded 4+4, 2+2, 1+1 (horizontal_sum_avx2, 119-126), the
 * division as a multiplication by 1 / scale and the packs saturation (-128 .. 127) of quantize_avx2 (1252-1289).
 * The seed mask comes from ChaCha8Rng::seed_from_u64(1234) in the reference (1845-1858; rand_chacha is not vendored): an input here.
 * The quantised vectors are then searched like any scaled i8 vectors: dot_i8_turboquant = dot * s1 * s2 (2072-2076),
 * euclidean_i8_turboquant = max(0, n1 + n2 - 2 dot_q) (2058-2069). */
uint32_t so_turboquant_dim(uint32_t n) { uint32_t d = 1; while (d < n) d <<= 1; return d; }
void so_turboquant_i8(const float* v, uint32_t n, const float* seed_mask, uint32_t dim, int avx2, int8_t* out, float* scale_out,
                      float* norm_out) {
  float* a = (float*)calloc(dim, sizeof(float));
  for (uint32_t i = 0; i < (n < dim ? n : dim); i++) a[i] = v[i];
  for (uint32_t i = 0; i < dim; i++) a[i] *= seed_mask[i];
  for (uint32_t h = 1; h < dim; h *= 2)
    for (uint32_t i = 0; i < dim; i += 2 * h)
      for (uint32_t j = i; j < i + h; j++) { const float x = a[j], y = a[j + h]; a[j] = x + y; a[j + h] = x - y; }
  const float nrm = sqrtf((float)dim);
  for (uint32_t i = 0; i < dim; i++) a[i] /= nrm;
  float sum_sq;
  if (avx2 && dim >= 8) {
    float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t i = 0; i + 8 <= dim; i += 8)
      for (int j = 0; j < 8; j++) { const float p = a[i + j] * a[i + j]; l[j] = l[j] + p; }
    const float x0 = l[4] + l[0], x1 = l[5] + l[1], x2 = l[6] + l[2], x3 = l[7] + l[3];
    const float y0 = x0 + x2, y1 = x1 + x3;
    sum_sq = y0 + y1;
  } else {
    sum_sq = 0.0f;
    for (uint32_t i = 0; i < dim; i++) sum_sq += a[i] * a[i];
  }
  const float sigma = sqrtf(sum_sq) / sqrtf((float)dim);
  float scale = sigma / 32.0f;
  if (!(scale > 1e-8f)) scale = 1e-8f;
  int32_t sq = 0;
  const float inv = 1.0f / scale;
  for (uint32_t i = 0; i < dim; i++) {
    int32_t q;
    if (avx2 && dim >= 16) {  /* x * (1 / scale), + copysign(0.5), truncate, saturating packs */
      const float s = a[i] * inv;
      const float adj = s + (signbit(s) ? -0.5f : 0.5f);
      q = (int32_t)adj;
      if (q > 127) q = 127;
      if (q < -128) q = -128;
    } else {
      float r = roundf(a[i] / scale);
      if (r < -127.0f) r = -127.0f;
      if (r > 127.0f) r = 127.0f;
      q = (int32_t)r;
    }
    out[i] = (int8_t)q;
    sq += q * q;
  }
  *scale_out = scale;
  *norm_out = (float)sq * scale * scale;
  free(a);
}

/* ---- Point (geo) facets: geo_search.rs ---- */
static uint64_t morton_part(uint32_t v) { /* encode_morton_64_bit, geo_search.rs:12-21 */
  uint64_t x = v;
  x = (x | (x << 32)) & 0x00000000ffffffffull;
  x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
  x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
  x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x << 2)) & 0x3333333333333333ull;
  x = (x | (x << 1)) & 0x5555555555555555ull;
  return x;
}
static uint32_t morton_unpart(uint64_t code) { /* decode_morton_64_bit, geo_search.rs:45-53 */
  uint64_t x = code & 0x5555555555555555ull;
  x = (x ^ (x >> 1)) & 0x3333333333333333ull;
  x = (x ^ (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x ^ (x >> 4)) & 0x00FF00FF00FF00FFull;
  x = (x ^ (x >> 8)) & 0x0000FFFF0000FFFFull;
  x = (x ^ (x >> 16)) & 0x00000000FFFFFFFFull;
  return (uint32_t)x;
}
static int32_t rust_f64_as_i32(double v) { /* `as i32`: toward zero, saturating, NaN -> 0 */
  if (v != v) return 0;
  if (v >= 2147483647.0) return INT32_MAX;
  if (v <= -2147483648.0) return INT32_MIN;
  return (int32_t)v;
}
uint64_t so_morton_encode(double lat, double lon) { /* geo_search.rs:27-41 */
  const uint32_t x = (uint32_t)rust_f64_as_i32(lat * 10000000.0), y = (uint32_t)rust_f64_as_i32(lon * 10000000.0);
  return (morton_part(y) << 1) | morton_part(x);
}
void so_morton_decode(uint64_t code, double* lat, double* lon) { /* geo_search.rs:58-79 */
  *lat = (double)(int32_t)morton_unpart(code) / 10000000.0;
  *lon = (double)(int32_t)morton_unpart(code >> 1) / 10000000.0;
}
#define SO_DEG2RAD 0.017453292519943295
static double so_earth_radius(int unit) { return unit == 1 ? 6371.0087714 : 3958.761315801475; }
void so_geo_distances(uint64_t n, const uint64_t* codes, double base_lat, double base_lon, int unit, double* out) {
  for (uint64_t i = 0; i < n; i++) {
    double lat, lon;
    so_morton_decode(codes[i], &lat, &lon);
    if (unit) { /* euclidian_distance(point1 = base, point2 = doc), geo_search.rs:115-124 */
      const double x = SO_DEG2RAD * (lon - base_lon) * cos(SO_DEG2RAD * (base_lat + lat) / 2.0);
      const double y = SO_DEG2RAD * (lat - base_lat);
      out[i] = so_earth_radius(unit) * sqrt(x * x + y * y);
    } else { /* simplified_distance(point1 = doc, point2 = base), geo_search.rs:82-87 */
      const double x = (base_lon - lon) * cos(SO_DEG2RAD * (lat + base_lat) / 2.0);
      const double y = base_lat - lat;
      out[i] = x * x + y * y;
    }
  }
}
void so_geo_morton_range(double lat, double lon, double distance, int unit, uint64_t out[2]) { /* geo_search.rs:128-144 */
  const double r = so_earth_radius(unit);
  const double lat_delta = distance / (SO_DEG2RAD * r);
  const double lon_delta = distance / (SO_DEG2RAD * r * cos(SO_DEG2RAD * lat));
  out[0] = so_morton_encode(lat - lat_delta, lon - lon_delta);
  out[1] = so_morton_encode(lat + lat_delta, lon + lon_delta);
}
