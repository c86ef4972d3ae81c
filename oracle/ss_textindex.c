/*
 * ss_textindex.c -- TEST INFRASTRUCTURE ONLY (like everything under oracle/).
 *
 * A text-shaped corpus and the reference's INDEXING path restated for it, so that the product's index.bin loader can be
 * rehearsed at configuration size on bytes laid out exactly as a SeekStorm shard holds them -- which oracle/ref_format.py
 * (pure Python, one posting at a time) cannot produce for a million docs.
 *
 *   corpus   : docs of lognormal length; every token a term RANK drawn Zipf(1) over the vocabulary (rank = floor(exp(u ln(V + 1)))
 *              - 1), except that a share of the tokens comes from the doc's TOPIC: a small set of mid-frequency terms shared by a
 *              run of consecutive doc ids (two cluster sizes).  Doc ids are therefore clustered per term -- Rle and Bitmap
 *              containers, uneven block maxima -- and real positions exist for phrases and n-grams.  Counter-based hashing
 *              (so_h): the same arguments give the same corpus.
 *   indexing : one indexed field.  Every token is a SingleTerm posting at its position; a pair / triple of consecutive FREQUENT
 *              terms is also indexed as an NgramFF / NgramFFF key at the position of its first word, its record carrying the
 *              positions count of every component term in the doc (tokenizer.rs:674-699, 751-782 with the default
 *              ngram_indexing = NgramFF | NgramFFF, index.rs:1422-1424; index_posting.rs:666-741).  "Frequent" = the
 *              n_frequent lowest ranks (the reference's frequent_hashset holds its frequent-word list, index.rs:1600-1640).
 *   writer   : index.bin as commit.rs:264-369 / commit_segment 467-552 write it: per 65 536-doc level the length bytes, the
 *              cumulative counters, the segment head table, then per segment the key heads (ascending key_hash,
 *              compress_postinglist.rs:339-409) and the key bodies -- position records stacked down from the pointer range,
 *              rank/position pointers (2 bytes up to the pivot, 3 after; embedded forms for SingleTerm postings of <= 4
 *              positions, index_posting.rs:445-660), doc-id container by the chooser (compress_postinglist.rs:256-332).
 *              The same rules as oracle/ref_format.py, which tests compare it with byte for byte.
 * Parity unpinned against the Rust binary (no toolchain here); the layout is pinned by the hand-assembled fixtures through
 * ref_format.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ss_oracle.h"

struct so_text {
  uint64_t seed, n_docs, n_tokens;
  uint32_t vocab, n_freq;
  uint32_t* doc_off;   /* [n_docs + 1] */
  uint32_t* tok;       /* term rank of every token */
  uint8_t* doclen;     /* int_to_byte4(token count) */
  /* keys: ids [0, vocab) = the single terms by rank (possibly without postings), then the n-gram keys */
  uint32_t n_keys;
  uint8_t* ncomp;      /* 1 | 2 | 3 */
  uint32_t* comp;      /* [n_keys][3] component term ranks */
  uint64_t* key_hash;
  uint64_t* key_off;   /* [n_keys + 1] first posting */
  uint64_t n_post;
  uint32_t* p_doc;     /* [n_post] */
  uint64_t* p_pos;     /* [n_post + 1] first position */
  uint16_t* pos;       /* [sum] */
  uint32_t* term_df;   /* [vocab] */
  /* several indexed fields (so_text_build_fields): a doc's tokens are cut into n_fields consecutive spans, positions restart in every
   * field, an n-gram never crosses a field boundary */
  uint32_t n_fields, longest_field;
  uint8_t* fdoclen;    /* [n_fields][n_docs] int_to_byte4(tokens of the field); NULL with one field */
  uint8_t* efld;       /* field of every entry of `pos` (NULL with one field) */
};

/* the span [a, b) (token indices) of field f of doc d: field 0 ("title") = the first max(1, L / 8) tokens, the last field ("tags") = the
 * last max(1, L / 8), the fields between share the rest evenly (the first of them takes the remainder); L >= 4 */
static void field_span(const so_text* T, uint64_t d, uint32_t f, uint32_t* a, uint32_t* b) {
  const uint32_t a0 = T->doc_off[d], L = T->doc_off[d + 1] - a0, F = T->n_fields;
  if (F == 1) { *a = a0; *b = a0 + L; return; }
  const uint32_t n0 = L / 8 ? L / 8 : 1, nl = F > 2 ? (L / 8 ? L / 8 : 1) : 0, mid = L - n0 - nl, nm = F > 2 ? F - 2 : 1;
  if (f == 0) { *a = a0; *b = a0 + n0; return; }
  if (F == 2) { *a = a0 + n0; *b = a0 + L; return; }
  if (f == F - 1) { *a = a0 + L - nl; *b = a0 + L; return; }
  const uint32_t each = mid / nm, extra = mid - each * nm, j = f - 1;  /* middle field j of nm */
  *a = a0 + n0 + j * each + (j ? extra : 0);
  *b = *a + each + (j == 0 ? extra : 0);
}

static uint64_t key_hash_of(uint64_t seed, uint32_t n, const uint32_t* c) {
  uint64_t h = so_h(seed ^ 0x6B65795F68617368ull, c[0] + 1u, n);
  for (uint32_t i = 1; i < n; i++) h = so_h(h, c[i] + 1u, i);
  return (h & ~7ull) | (n == 1 ? 0ull : n == 2 ? 1ull : 4ull);  /* NgramType: SingleTerm 0, NgramFF 1, NgramFFF 4 (index.rs:1854-1872) */
}

/* open-addressing table (a, b, c) -> n-gram key id */
typedef struct { uint64_t* k; uint32_t* v; uint64_t mask; } ng_tab;
static uint64_t ng_pack(uint32_t n, const uint32_t* c) { return ((uint64_t)n << 60) | ((uint64_t)c[0] << 40) | ((uint64_t)c[1] << 20) | (n == 3 ? c[2] : 0u); }

void so_text_free(so_text* t) {
  if (!t) return;
  free(t->doc_off); free(t->tok); free(t->doclen); free(t->ncomp); free(t->comp); free(t->key_hash); free(t->key_off);
  free(t->p_doc); free(t->p_pos); free(t->pos); free(t->term_df); free(t->fdoclen); free(t->efld); free(t);
}

/* n_frequent < 2^20 (ranks packed in 20 bits for the n-gram table); ngrams: bit 0 = NgramFF, bit 3 = NgramFFF (NgramSet, index.rs:1840-1850) */
so_text* so_text_build(uint64_t seed, uint64_t n_docs, uint32_t vocab, uint32_t n_frequent, int ngrams, double topic_share, double mean_len) {
  return so_text_build_fields(seed, n_docs, vocab, n_frequent, ngrams, topic_share, mean_len, 1, 0);
}
so_text* so_text_build_fields(uint64_t seed, uint64_t n_docs, uint32_t vocab, uint32_t n_frequent, int ngrams, double topic_share, double mean_len,
                              uint32_t n_fields, uint32_t longest_field) {
  if (n_docs == 0 || n_docs > 0xFFFFFFFFull || vocab < 64 || n_frequent >= (1u << 20) || n_fields == 0 || n_fields > 4 || longest_field >= n_fields) return NULL;
  so_text* T = (so_text*)calloc(1, sizeof(so_text));
  T->seed = seed; T->n_docs = n_docs; T->vocab = vocab; T->n_freq = n_frequent; T->n_fields = n_fields; T->longest_field = longest_field;
  T->doc_off = (uint32_t*)malloc((n_docs + 1) * sizeof(uint32_t));
  T->doclen = (uint8_t*)malloc(n_docs);
  /* lengths: clamp(round(exp(ln mean_len + 0.55 z)), 4, 1500), z ~ N(0,1) by Box-Muller on two hash words */
  const double ln_mean = log(mean_len > 1.0 ? mean_len : 100.0);
  uint64_t total = 0;
  for (uint64_t d = 0; d < n_docs; d++) {
    const uint64_t h1 = so_h(seed, d, 0xA1), h2 = so_h(seed, d, 0xA2);
    const double u1 = ((double)(h1 >> 11) + 1.0) / 9007199254740993.0, u2 = (double)(h2 >> 11) / 9007199254740992.0;
    const double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    double L = floor(exp(ln_mean + 0.55 * z) + 0.5);
    if (L < 4) L = 4;
    if (L > 1500) L = 1500;
    T->doc_off[d] = (uint32_t)total;
    total += (uint64_t)L;
    T->doclen[d] = so_int_to_byte4((uint32_t)L);
    if (total > 0xFFFFFFF0ull) { so_text_free(T); return NULL; }
  }
  T->doc_off[n_docs] = (uint32_t)total;
  T->n_tokens = total;
  T->tok = (uint32_t*)malloc((total ? total : 1) * sizeof(uint32_t));
  const double lnv = log((double)vocab + 1.0);
  const uint64_t topic_cut = (uint64_t)(topic_share * 18446744073709551615.0);
  for (uint64_t d = 0; d < n_docs; d++) {
    /* the doc's topic: 24 terms from ranks [64, 32 768) chosen by the doc's cluster; half of the docs cluster by 1024 ids, half by 128 */
    const uint64_t cl = (so_h(seed, d >> 10, 0xC0) & 1u) ? (d >> 10) * 2u + 1u : (d >> 7) * 2u;
    for (uint32_t i = T->doc_off[d]; i < T->doc_off[d + 1]; i++) {
      const uint64_t h = so_h(seed, d, 0x100u + (i - T->doc_off[d]));
      uint32_t r;
      if (so_splitmix64(h) < topic_cut) {
        const uint64_t j = so_h(seed ^ 0x746F706963ull, cl, h % 24u);
        r = 64u + (uint32_t)(j % 32704u);
        if (r >= vocab) r = (uint32_t)(j % vocab);
      } else {
        const double u = (double)(h >> 11) / 9007199254740992.0;
        double x = floor(exp(u * lnv));
        r = x < 1.0 ? 0u : (uint32_t)x - 1u;
        if (r >= vocab) r = vocab - 1u;
      }
      T->tok[i] = r;
    }
  }
  if (n_fields > 1) {
    T->fdoclen = (uint8_t*)malloc((size_t)n_fields * n_docs);
    for (uint64_t d = 0; d < n_docs; d++)
      for (uint32_t f = 0; f < n_fields; f++) {
        uint32_t a, b;
        field_span(T, d, f, &a, &b);
        T->fdoclen[(size_t)f * n_docs + d] = so_int_to_byte4(b - a);
      }
  }
  /* ---- n-gram keys: discover the distinct ones */
  ng_tab G; G.mask = (1ull << 22) - 1; G.k = (uint64_t*)calloc(G.mask + 1, 8); G.v = (uint32_t*)malloc((G.mask + 1) * 4);
  uint32_t n_ng = 0, ng_cap = 1u << 16;
  uint32_t* ng_comp = (uint32_t*)malloc((size_t)ng_cap * 3 * 4);
  uint8_t* ng_n = (uint8_t*)malloc(ng_cap);
#define NG_LOOKUP(N_, C_, OUT_)                                                                        \
  do {                                                                                                 \
    const uint64_t pk_ = ng_pack(N_, C_);                                                              \
    uint64_t s_ = so_splitmix64(pk_) & G.mask;                                                         \
    while (G.k[s_] && G.k[s_] != pk_) s_ = (s_ + 1) & G.mask;                                          \
    if (!G.k[s_]) {                                                                                    \
      if (n_ng * 2ull > G.mask) { /* grow */                                                           \
        ng_tab N2; N2.mask = G.mask * 2 + 1; N2.k = (uint64_t*)calloc(N2.mask + 1, 8); N2.v = (uint32_t*)malloc((N2.mask + 1) * 4); \
        for (uint64_t q_ = 0; q_ <= G.mask; q_++) if (G.k[q_]) { uint64_t z_ = so_splitmix64(G.k[q_]) & N2.mask; while (N2.k[z_]) z_ = (z_ + 1) & N2.mask; N2.k[z_] = G.k[q_]; N2.v[z_] = G.v[q_]; } \
        free(G.k); free(G.v); G = N2;                                                                  \
        s_ = so_splitmix64(pk_) & G.mask;                                                              \
        while (G.k[s_]) s_ = (s_ + 1) & G.mask;                                                        \
      }                                                                                                \
      if (n_ng == ng_cap) { ng_cap *= 2; ng_comp = (uint32_t*)realloc(ng_comp, (size_t)ng_cap * 3 * 4); ng_n = (uint8_t*)realloc(ng_n, ng_cap); } \
      G.k[s_] = pk_; G.v[s_] = n_ng;                                                                   \
      ng_n[n_ng] = (uint8_t)(N_); ng_comp[3 * n_ng] = (C_)[0]; ng_comp[3 * n_ng + 1] = (C_)[1]; ng_comp[3 * n_ng + 2] = (N_) == 3 ? (C_)[2] : 0u; \
      n_ng++;                                                                                          \
    }                                                                                                  \
    (OUT_) = G.v[s_];                                                                                  \
  } while (0)
  /* pass A: count entries per key (single terms: tokens; n-grams: occurrences) */
  uint64_t* cnt = (uint64_t*)calloc((size_t)vocab + 1, 8);
  for (uint64_t i = 0; i < total; i++) cnt[T->tok[i]]++;
  uint64_t ng_occ = 0;
  uint32_t* occ_key = NULL;   /* n-gram occurrences in token order: key id (relative), doc, pos */
  uint32_t* occ_doc = NULL; uint16_t* occ_pos = NULL; uint8_t* occ_fld = NULL;
  uint64_t occ_cap = 0;
  if (ngrams & 9) {
    for (uint64_t d = 0; d < n_docs; d++)
     for (uint32_t fl = 0; fl < n_fields; fl++) {  /* an n-gram stands inside ONE field: positions count from the field's first token */
      uint32_t a, b;
      field_span(T, d, fl, &a, &b);
      for (uint32_t i = a + 1; i < b; i++) {
        const uint32_t t0 = T->tok[i], t1 = T->tok[i - 1];
        if (t0 >= n_frequent || t1 >= n_frequent) continue;
        for (int tri = 0; tri < 2; tri++) {
          if (tri == 0 && !(ngrams & 1)) continue;
          if (tri == 1 && (!(ngrams & 8) || i < a + 2 || T->tok[i - 2] >= n_frequent)) continue;
          uint32_t c[3];
          uint32_t id;
          if (tri) { c[0] = T->tok[i - 2]; c[1] = t1; c[2] = t0; } else { c[0] = t1; c[1] = t0; c[2] = 0; }
          NG_LOOKUP(tri ? 3u : 2u, c, id);
          if (ng_occ == occ_cap) {
            occ_cap = occ_cap ? occ_cap * 2 : (1u << 20);
            occ_key = (uint32_t*)realloc(occ_key, occ_cap * 4); occ_doc = (uint32_t*)realloc(occ_doc, occ_cap * 4); occ_pos = (uint16_t*)realloc(occ_pos, occ_cap * 2);
            occ_fld = (uint8_t*)realloc(occ_fld, occ_cap);
          }
          occ_key[ng_occ] = id; occ_doc[ng_occ] = (uint32_t)d; occ_pos[ng_occ] = (uint16_t)(i - a - (tri ? 2u : 1u));  /* the place of the FIRST word */
          occ_fld[ng_occ] = (uint8_t)fl;
          ng_occ++;
        }
      }
     }
  }
  free(G.k); free(G.v);
  /* ---- keys */
  T->n_keys = vocab + n_ng;
  T->ncomp = (uint8_t*)malloc(T->n_keys);
  T->comp = (uint32_t*)calloc((size_t)T->n_keys * 3, 4);
  T->key_hash = (uint64_t*)malloc((size_t)T->n_keys * 8);
  for (uint32_t r = 0; r < vocab; r++) { T->ncomp[r] = 1; T->comp[3 * (size_t)r] = r; T->key_hash[r] = key_hash_of(seed, 1, &r); }
  for (uint32_t g = 0; g < n_ng; g++) {
    const uint32_t k = vocab + g;
    T->ncomp[k] = ng_n[g];
    memcpy(T->comp + 3 * (size_t)k, ng_comp + 3 * (size_t)g, 12);
    T->key_hash[k] = key_hash_of(seed, ng_n[g], ng_comp + 3 * (size_t)g);
  }
  free(ng_comp); free(ng_n);
  /* ---- entries (key, doc, pos) sorted by key (stable: doc / pos order kept), then grouped into postings */
  const uint64_t n_ent = total + ng_occ;
  uint64_t* eoff = (uint64_t*)calloc((size_t)T->n_keys + 1, 8);
  for (uint32_t r = 0; r < vocab; r++) eoff[r + 1] = cnt[r];
  for (uint64_t i = 0; i < ng_occ; i++) eoff[vocab + occ_key[i] + 1]++;
  for (uint32_t k = 0; k < T->n_keys; k++) eoff[k + 1] += eoff[k];
  uint32_t* e_doc = (uint32_t*)malloc((n_ent ? n_ent : 1) * 4);
  uint16_t* e_pos = (uint16_t*)malloc((n_ent ? n_ent : 1) * 2);
  uint8_t* e_fld = n_fields > 1 ? (uint8_t*)malloc(n_ent ? n_ent : 1) : NULL;
  uint64_t* cur = (uint64_t*)malloc((size_t)T->n_keys * 8);
  memcpy(cur, eoff, (size_t)T->n_keys * 8);
  for (uint64_t d = 0; d < n_docs; d++)
    for (uint32_t fl = 0; fl < n_fields; fl++) {
      uint32_t a, b;
      field_span(T, d, fl, &a, &b);
      for (uint32_t i = a; i < b; i++) {
        const uint64_t w = cur[T->tok[i]]++;
        e_doc[w] = (uint32_t)d; e_pos[w] = (uint16_t)(i - a);
        if (e_fld) e_fld[w] = (uint8_t)fl;
      }
    }
  for (uint64_t i = 0; i < ng_occ; i++) {
    const uint64_t w = cur[vocab + occ_key[i]]++;
    e_doc[w] = occ_doc[i]; e_pos[w] = occ_pos[i];
    if (e_fld) e_fld[w] = occ_fld[i];
  }
  T->efld = e_fld;
  free(cur); free(cnt); free(occ_key); free(occ_doc); free(occ_pos); free(occ_fld);
  /* postings = runs of equal doc inside a key */
  T->key_off = (uint64_t*)malloc(((size_t)T->n_keys + 1) * 8);
  uint64_t np = 0;
  for (uint32_t k = 0; k < T->n_keys; k++)
    for (uint64_t e = eoff[k]; e < eoff[k + 1]; e++) np += (e == eoff[k] || e_doc[e] != e_doc[e - 1]);
  T->n_post = np;
  T->p_doc = (uint32_t*)malloc((np ? np : 1) * 4);
  T->p_pos = (uint64_t*)malloc((np + 1) * 8);
  T->pos = e_pos;  /* the entries' positions are the postings' positions, in order */
  T->term_df = (uint32_t*)calloc(vocab, 4);
  np = 0;
  for (uint32_t k = 0; k < T->n_keys; k++) {
    T->key_off[k] = np;
    for (uint64_t e = eoff[k]; e < eoff[k + 1]; e++)
      if (e == eoff[k] || e_doc[e] != e_doc[e - 1]) { T->p_doc[np] = e_doc[e]; T->p_pos[np] = e; np++; }
    if (k < vocab) T->term_df[k] = (uint32_t)(np - T->key_off[k]);
  }
  T->key_off[T->n_keys] = np;
  T->p_pos[np] = n_ent;
  free(e_doc); free(eoff);
  return T;
}

void so_text_info(const so_text* T, uint64_t* n_tokens, uint32_t* n_keys, uint32_t* n_keys_nonempty, uint64_t* n_postings, uint32_t* n_ngram_keys) {
  uint32_t ne = 0;
  for (uint32_t k = 0; k < T->n_keys; k++) ne += T->key_off[k + 1] > T->key_off[k];
  if (n_tokens) *n_tokens = T->n_tokens;
  if (n_keys) *n_keys = T->n_keys;
  if (n_keys_nonempty) *n_keys_nonempty = ne;
  if (n_postings) *n_postings = T->n_post;
  if (n_ngram_keys) *n_ngram_keys = T->n_keys - T->vocab;
}
const uint8_t* so_text_doclen(const so_text* T) { return T->doclen; }
/* tokens of doc d (term ranks); returns the count */
uint32_t so_text_doc_tokens(const so_text* T, uint64_t d, uint32_t cap, uint32_t* out) {
  const uint32_t n = T->doc_off[d + 1] - T->doc_off[d];
  for (uint32_t i = 0; i < n && i < cap; i++) out[i] = T->tok[T->doc_off[d] + i];
  return n;
}
/* key id of an n-gram (2 or 3 component ranks), 0xFFFFFFFF if the corpus holds none; single terms: id = rank */
uint32_t so_text_ngram_key(const so_text* T, uint32_t n, const uint32_t* c) {
  const uint64_t h = key_hash_of(T->seed, n, c);
  for (uint32_t k = T->vocab; k < T->n_keys; k++)
    if (T->key_hash[k] == h) return k;
  return 0xFFFFFFFFu;
}
uint64_t so_text_key_hash(const so_text* T, uint32_t key) { return T->key_hash[key]; }
uint64_t so_text_key_df(const so_text* T, uint32_t key) { return T->key_off[key + 1] - T->key_off[key]; }
/* the posting list of a key as the oracle's shard model wants it: docs, per posting the positions count (for an n-gram key: the
 * key's own) and, for component c of an n-gram key, the component term's tf in the doc; positions appended (absolute, ascending) */
uint64_t so_text_key_postings(const so_text* T, uint32_t key, uint32_t component, uint32_t* docs, uint16_t* tfs, uint16_t* counts,
                              uint16_t* positions, uint64_t pos_cap, uint64_t* n_pos_out) {
  const uint64_t a = T->key_off[key], b = T->key_off[key + 1];
  uint64_t np = 0;
  for (uint64_t p = a; p < b; p++) {
    const uint64_t c = T->p_pos[p + 1] - T->p_pos[p];
    if (docs) docs[p - a] = T->p_doc[p];
    if (counts) counts[p - a] = (uint16_t)c;
    if (tfs) {
      if (T->ncomp[key] == 1) tfs[p - a] = (uint16_t)c;
      else {  /* the component term's tf in this doc: count its tokens */
        const uint32_t r = T->comp[3 * (size_t)key + component], d = T->p_doc[p];
        uint32_t tf = 0;
        for (uint32_t i = T->doc_off[d]; i < T->doc_off[d + 1]; i++) tf += T->tok[i] == r;
        tfs[p - a] = (uint16_t)tf;
      }
    }
    if (positions) for (uint64_t x = 0; x < c && np + x < pos_cap; x++) positions[np + x] = T->pos[T->p_pos[p] + x];
    np += c;
  }
  if (n_pos_out) *n_pos_out = np;
  return b - a;
}

const uint8_t* so_text_doclen_fields(const so_text* T) { return T->n_fields > 1 ? T->fdoclen : T->doclen; }
uint32_t so_text_fields(const so_text* T, uint32_t* longest_field) { if (longest_field) *longest_field = T->longest_field; return T->n_fields; }
/* tokens of field f of doc d */
uint32_t so_text_doc_field_tokens(const so_text* T, uint64_t d, uint32_t f, uint32_t cap, uint32_t* out) {
  uint32_t a, b;
  field_span(T, d, f, &a, &b);
  for (uint32_t i = a; i < b && i - a < cap; i++) out[i - a] = T->tok[i];
  return b - a;
}
/* Several indexed fields: the (doc, field) ENTRIES of a key as the oracle's multi-field model wants them (so_search_fields_*_items):
 * a single term: one entry per field that holds it, tf = count = its positions there; component c of an n-gram key: one entry per
 * field of the doc in which the COMPONENT TERM occurs (its tf there), count = the key's own positions in that field for c == 0 (0
 * where the key does not stand in the field, and for the other components), positions = the key's, field after field.
 * Returns the number of entries (call with NULL arrays to size them). */
uint64_t so_text_key_entries(const so_text* T, uint32_t key, uint32_t component, uint32_t* docs, uint8_t* fields, uint16_t* tfs, uint16_t* counts,
                             uint16_t* positions, uint64_t pos_cap, uint64_t* n_pos_out) {
  const uint64_t a = T->key_off[key], b = T->key_off[key + 1];
  const uint32_t F = T->n_fields;
  uint64_t ne = 0, np = 0;
  for (uint64_t p = a; p < b; p++) {
    const uint32_t d = T->p_doc[p];
    const uint64_t e0 = T->p_pos[p], e1 = T->p_pos[p + 1];
    for (uint32_t f = 0; f < F; f++) {
      uint32_t own = 0;  /* the key's positions in field f */
      uint64_t first = e1;
      for (uint64_t e = e0; e < e1; e++)
        if ((T->efld ? T->efld[e] : 0u) == f) { if (!own) first = e; own++; }
      uint32_t tf = own;
      if (T->ncomp[key] > 1) {
        const uint32_t r = T->comp[3 * (size_t)key + component];
        uint32_t fa, fb;
        field_span(T, d, f, &fa, &fb);
        tf = 0;
        for (uint32_t i = fa; i < fb; i++) tf += T->tok[i] == r;
        if (component != 0) own = 0;
      }
      if (!tf) continue;
      if (docs) docs[ne] = d;
      if (fields) fields[ne] = (uint8_t)f;
      if (tfs) tfs[ne] = (uint16_t)tf;
      if (counts) counts[ne] = (uint16_t)own;
      if (positions) for (uint32_t x = 0; x < own && np + x < pos_cap; x++) positions[np + x] = T->pos[first + x];
      np += own;
      ne++;
    }
  }
  if (n_pos_out) *n_pos_out = np;
  return ne;
}

/* ================================================================== index.bin writer */
typedef struct { uint8_t* p; uint64_t n, cap; } buf;
static void b_need(buf* b, uint64_t more) {
  if (b->n + more <= b->cap) return;
  while (b->n + more > b->cap) b->cap = b->cap ? b->cap * 2 : 4096;
  b->p = (uint8_t*)realloc(b->p, b->cap);
}
static void b_put(buf* b, const void* src, uint64_t n) { b_need(b, n); memcpy(b->p + b->n, src, n); b->n += n; }
static void b_u8(buf* b, uint32_t v) { b_need(b, 1); b->p[b->n++] = (uint8_t)v; }
static void b_le(buf* b, uint64_t v, int bytes) { b_need(b, (uint64_t)bytes); for (int i = 0; i < bytes; i++) b->p[b->n++] = (uint8_t)(v >> (8 * i)); }
/* count VINT (write_field_vec, one indexed field: index_posting.rs:858-873) */
static void b_vint(buf* b, uint32_t v) {
  if (v < 128) b_u8(b, v | 0x80);
  else if (v < 16384) { b_u8(b, v >> 7); b_u8(b, (v & 0x7F) | 0x80); }
  else { b_u8(b, v >> 14); b_u8(b, (v >> 7) & 0x7F); b_u8(b, (v & 0x7F) | 0x80); }
}
/* position VINT (compress_positions, compress_postinglist.rs:948-976; the three-byte form keeps bit 13 twice) */
static void b_posvint(buf* b, uint32_t d) {
  if (d < 128) b_u8(b, d | 0x80);
  else if (d < 16384) { b_u8(b, (d >> 7) & 0x7F); b_u8(b, (d & 0x7F) | 0x80); }
  else { b_u8(b, (d >> 13) & 0x7F); b_u8(b, (d >> 7) & 0x7F); b_u8(b, (d & 0x7F) | 0x80); }
}
static uint32_t bitlen(uint32_t v) { uint32_t n = 0; while (v) { n++; v >>= 1; } return n; }
static int embeddable(uint32_t n, const uint32_t* dl, int psize) {  /* index_posting.rs:447-471 */
  if (n == 0 || n > 4) return 0;
  uint32_t b[4];
  for (uint32_t i = 0; i < n; i++) b[i] = bitlen(dl[i]);
  if (psize == 2) return (n == 1 && b[0] <= 14) || (n == 2 && b[0] <= 7 && b[1] <= 7);
  return (n == 1 && b[0] <= 21) || (n == 2 && b[0] <= 10 && b[1] <= 11) || (n == 3 && b[0] <= 7 && b[1] <= 7 && b[2] <= 7) ||
         (n == 4 && b[0] <= 5 && b[1] <= 5 && b[2] <= 5 && b[3] <= 6);
}
static void embed(buf* out, uint32_t n, const uint32_t* dl, int psize) {  /* index_posting.rs:592-640 */
  uint32_t remaining = (uint32_t)psize * 8u - (psize == 2 ? 0u : 1u) - 2u, data = 0;
  for (uint32_t i = 0; i < n; i++) { const uint32_t w = remaining / (n - i); remaining -= w; data = (data << w) | dl[i]; }
  if (psize == 2) { b_u8(out, data & 0xFF); b_u8(out, ((data >> 8) | 0x80 | ((n - 1) << 6)) & 0xFF); }
  else { b_u8(out, data & 0xFF); b_u8(out, (data >> 8) & 0xFF); b_u8(out, ((data >> 16) | 0x80 | ((n - 1) << 5)) & 0xFF); }
}

static void container_of(const so_text* T, uint64_t a, uint32_t n, buf* bodies, uint32_t* ctype_out);
/* one key's postings [a, b) of one level -> body appended to `bodies`; returns head fields */
static void encode_key_body(const so_text* T, uint32_t key, uint64_t a, uint64_t b, uint32_t positions_limit, buf* bodies, buf* recs /* scratch */,
                            buf* ptrs /* scratch */, uint32_t* ctp_out, uint32_t* pivot_out) {
  const uint32_t n = (uint32_t)(b - a), nc = T->ncomp[key];
  const uint64_t base = bodies->n;
  uint32_t size_positions = 0, pivot = 0;
  int three = 0;
  recs->n = 0; ptrs->n = 0;
  /* records are stacked DOWNWARD: collected in order here with their sizes, then written in reverse */
  uint32_t* rec_end = (uint32_t*)malloc((size_t)n * 4);
  uint32_t n_rec = 0;
  for (uint32_t r = 0; r < n; r++) {
    const uint64_t p = a + r;
    const uint32_t c = (uint32_t)(T->p_pos[p + 1] - T->p_pos[p]);
    const uint16_t* ps = T->pos + T->p_pos[p];
    int psize;
    if (!three && size_positions < positions_limit && r < 65535u) { pivot = r + 1; psize = 2; }   /* index_posting.rs:193-199 */
    else { psize = 3; three = 1; }
    uint32_t dl[4];
    if (nc == 1 && c <= 4) {
      for (uint32_t i = 0; i < c; i++) dl[i] = i == 0 ? ps[0] : (uint32_t)ps[i] - ps[i - 1] - 1u;
      if (embeddable(c, dl, psize)) { embed(ptrs, c, dl, psize); continue; }
    }
    const uint64_t r0 = recs->n;
    if (nc > 1) {  /* the component terms' counts first (index_posting.rs:666-722) */
      const uint32_t d = T->p_doc[p];
      for (uint32_t ci = 0; ci < nc; ci++) {
        const uint32_t rk = T->comp[3 * (size_t)key + ci];
        uint32_t tf = 0;
        for (uint32_t i = T->doc_off[d]; i < T->doc_off[d + 1]; i++) tf += T->tok[i] == rk;
        b_vint(recs, tf);
      }
    }
    b_vint(recs, c);
    for (uint32_t i = 0; i < c; i++) b_posvint(recs, i == 0 ? ps[0] : (uint32_t)ps[i] - ps[i - 1] - 1u);
    const uint32_t len = (uint32_t)(recs->n - r0);
    if (psize == 2 && size_positions + len >= positions_limit) { psize = 3; pivot = r; three = 1; }  /* index_posting.rs:579-587 */
    size_positions += len;
    rec_end[n_rec++] = (uint32_t)recs->n;
    if (psize == 2) { b_u8(ptrs, size_positions & 255); b_u8(ptrs, (size_positions >> 8) & 127); }
    else { b_u8(ptrs, size_positions & 255); b_u8(ptrs, (size_positions >> 8) & 255); b_u8(ptrs, (size_positions >> 16) & 127); }
  }
  for (uint32_t i = n_rec; i > 0; i--) {  /* reversed(records) */
    const uint32_t s = i > 1 ? rec_end[i - 2] : 0u, e = rec_end[i - 1];
    b_put(bodies, recs->p + s, e - s);
  }
  free(rec_end);
  b_put(bodies, ptrs->p, ptrs->n);
  uint32_t ctype;
  container_of(T, a, n, bodies, &ctype);
  *ctp_out = (ctype << 30) | (uint32_t)(base + size_positions);
  *pivot_out = pivot;
}
/* doc-id container: chooser compress_postinglist.rs:256-332, writers :694 / :759 / :832 */
static void container_of(const so_text* T, uint64_t a, uint32_t n, buf* bodies, uint32_t* ctype_out) {
  uint32_t runs = 1;
  for (uint32_t r = 1; r < n; r++) runs += (T->p_doc[a + r] & 0xFFFFu) != (T->p_doc[a + r - 1] & 0xFFFFu) + 1u;
  const uint32_t thr = n < 4096 ? n / 2 : 2048;
  uint32_t ctype;
  if (thr > 0 && runs - 1 < thr) {
    ctype = 3;
    b_le(bodies, runs, 2);
    uint32_t s = 0;
    for (uint32_t r = 1; r <= n; r++)
      if (r == n || (T->p_doc[a + r] & 0xFFFFu) != (T->p_doc[a + r - 1] & 0xFFFFu) + 1u) {
        b_le(bodies, T->p_doc[a + s] & 0xFFFFu, 2);
        b_le(bodies, r - s - 1, 2);
        s = r;
      }
  } else if (n < 4096) {
    ctype = 1;
    for (uint32_t r = 0; r < n; r++) b_le(bodies, T->p_doc[a + r] & 0xFFFFu, 2);
  } else {
    ctype = 2;
    b_need(bodies, 8192);
    memset(bodies->p + bodies->n, 0, 8192);
    for (uint32_t r = 0; r < n; r++) { const uint32_t d = T->p_doc[a + r] & 0xFFFFu; bodies->p[bodies->n + (d >> 3)] |= (uint8_t)(1u << (d & 7)); }
    bodies->n += 8192;
  }
  *ctype_out = ctype;
}

/* ---- several indexed fields (index_posting.rs:433-940; the rules of oracle/ref_format.py encode_key_body_fields, byte for byte) */
static uint32_t field_id_bits(uint32_t n_fields) { return bitlen(n_fields - 1u); }  /* index.rs:2569-2570 */
/* [(field id, positions_count)] in front of a position record (index_posting.rs:846-940) */
static void write_field_vec(buf* out, uint32_t n, const uint32_t* fid, const uint32_t* cnt, int only_longest, uint32_t id_bits) {
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t c = cnt[i];
    if (only_longest) {
      if (c < 64) b_u8(out, c | 0xC0);
      else if (c < 8192) { b_u8(out, (c >> 7) | 0x40); b_u8(out, (c & 0x7F) | 0x80); }
      else { b_u8(out, (c >> 14) | 0x40); b_u8(out, (c >> 7) & 0x7F); b_u8(out, (c & 0x7F) | 0x80); }
      continue;
    }
    const uint32_t stop = i == n - 1 ? (i == 0 ? 0x20u : 0x40u) : 0u;  /* FIELD_STOP_BIT_1 / _2 (index.rs:112-113) */
    const uint32_t v = (c << id_bits) | fid[i];
    const uint32_t meta = (i == 0 ? 1u : 0u) + bitlen(c) + id_bits;
    if (meta <= 6) b_u8(out, stop | v | 0x80);
    else if (meta <= 13) { b_u8(out, stop | (v >> 7)); b_u8(out, (v & 0x7F) | 0x80); }
    else { b_u8(out, stop | (v >> 14)); b_u8(out, (v >> 7) & 0x7F); b_u8(out, (v & 0x7F) | 0x80); }
  }
}
/* nf non-empty fields (ids fid[], counts cnt[]), their delta positions concatenated in dl[n] (index_posting.rs:472-562) */
static int embeddable_fields(uint32_t n, const uint32_t* dl, uint32_t nf, int only_longest, uint32_t id_bits, int psize) {
  uint32_t b[4];
  if (n == 0 || n > 4) return 0;
  for (uint32_t i = 0; i < n; i++) b[i] = bitlen(dl[i]);
  if (only_longest) {
    if (psize == 2) return (n == 1 && b[0] <= 13) || (n == 2 && b[0] <= 6 && b[1] <= 7);
    uint32_t mx = 0;
    for (uint32_t i = 0; i < n; i++) mx = b[i] > mx ? b[i] : mx;
    return (n == 1 && b[0] <= 20) || (n == 2 && b[0] <= 10 && b[1] <= 10) || (n == 3 && b[0] <= 6 && b[1] <= 7 && b[2] <= 7) || (n == 4 && mx <= 5);
  }
  const uint32_t used = nf * id_bits, bits = psize == 2 ? 12u : 19u;
  if (used >= bits) return 0;
  const uint32_t rem = bits - used;
  if (n == 1) return b[0] <= rem;
  if (n == 2) return b[0] <= rem / 2 && b[1] <= rem - rem / 2;
  if (n == 3 && (psize == 3 || nf == 1)) return b[0] <= rem / 3 && b[1] <= (rem - rem / 3) / 2 && b[2] <= rem - (rem - rem / 3) / 2 - rem / 3;
  if (n == 4 && psize == 3 && nf == 1) {
    const uint32_t b2 = (rem - rem / 4) / 3, b3 = (rem - b2 - rem / 4) / 2;
    return b[0] <= rem / 4 && b[1] <= b2 && b[2] <= b3 && b[3] <= rem - rem / 4 - b2 - b3;
  }
  return 0;
}
static void embed_fields(buf* out, uint32_t n, const uint32_t* dl, uint32_t nf, const uint32_t* fid, const uint32_t* cnt, int only_longest,
                         uint32_t id_bits, int psize) {  /* index_posting.rs:592-660 */
  uint32_t data = 0;
  if (!only_longest)
    for (uint32_t i = 0; i < nf; i++) data = (data << id_bits) | fid[i];
  uint32_t remaining = (uint32_t)psize * 8u - (psize == 2 ? 0u : 1u) - (only_longest ? 3u : 4u + nf * id_bits);
  for (uint32_t i = 0; i < n; i++) { const uint32_t w = remaining / (n - i); remaining -= w; data = (data << w) | dl[i]; }
  if (psize == 2) {
    uint32_t hi;
    if (only_longest) hi = (data >> 8) | 0xC0 | ((n - 1) << 5);
    else if (nf == 1) hi = (data >> 8) | 0x80 | ((n - 1) << 4);
    else hi = (data >> 8) | 0xB0;
    b_u8(out, data & 0xFF); b_u8(out, hi & 0xFF);
    return;
  }
  uint32_t top;
  if (only_longest) top = (data >> 16) | 0xC0 | ((n - 1) << 4);
  else top = (data >> 16) | 0x80 | (nf == 1 ? ((n - 1) << 3) : nf == 3 ? 0x38u : (cnt[0] == 1 && cnt[1] == 1) ? 0x20u : (cnt[0] == 1 && cnt[1] == 2) ? 0x28u : 0x30u);
  b_u8(out, data & 0xFF); b_u8(out, (data >> 8) & 0xFF); b_u8(out, top & 0xFF);
}
static void container_of(const so_text* T, uint64_t a, uint32_t n, buf* bodies, uint32_t* ctype_out);

static void encode_key_body_fields(const so_text* T, uint32_t key, uint64_t a, uint64_t b, uint32_t positions_limit, buf* bodies, buf* recs, buf* ptrs,
                                   uint32_t* ctp_out, uint32_t* pivot_out) {
  const uint32_t n = (uint32_t)(b - a), nc = T->ncomp[key], F = T->n_fields, id_bits = field_id_bits(F);
  const uint64_t base = bodies->n;
  uint32_t size_positions = 0, pivot = 0;
  int three = 0;
  recs->n = 0; ptrs->n = 0;
  uint32_t* rec_end = (uint32_t*)malloc((size_t)n * 4);
  uint32_t n_rec = 0;
  uint32_t* dl = NULL;
  uint32_t dl_cap = 0;
  for (uint32_t r = 0; r < n; r++) {
    const uint64_t p = a + r, e0 = T->p_pos[p], e1 = T->p_pos[p + 1];
    const uint32_t total = (uint32_t)(e1 - e0);
    /* the non-empty fields of the posting, ascending, and the delta positions field after field */
    uint32_t fid[8], cnt[8], nf = 0;
    if (total > dl_cap) { dl_cap = total * 2; dl = (uint32_t*)realloc(dl, (size_t)dl_cap * 4); }
    for (uint64_t e = e0; e < e1; e++) {
      const uint32_t f = T->efld[e];
      if (nf == 0 || fid[nf - 1] != f) { fid[nf] = f; cnt[nf] = 0; nf++; }
      dl[e - e0] = cnt[nf - 1] == 0 ? T->pos[e] : (uint32_t)T->pos[e] - T->pos[e - 1] - 1u;
      cnt[nf - 1]++;
    }
    const int only_longest = nf == 1 && fid[0] == T->longest_field;
    int psize;
    if (!three && size_positions < positions_limit && r < 65535u) { pivot = r + 1; psize = 2; }
    else { psize = 3; three = 1; }
    if (nc == 1 && total <= 4 && embeddable_fields(total, dl, nf, only_longest, id_bits, psize)) {
      embed_fields(ptrs, total, dl, nf, fid, cnt, only_longest, id_bits, psize);
      continue;
    }
    const uint64_t r0 = recs->n;
    if (nc > 1) {  /* the component terms' field vectors first (index_posting.rs:664-722) */
      const uint32_t d = T->p_doc[p];
      for (uint32_t ci = 0; ci < nc; ci++) {
        const uint32_t rk = T->comp[3 * (size_t)key + ci];
        uint32_t cf[8], cc[8], ncf = 0;
        for (uint32_t f = 0; f < F; f++) {
          uint32_t fa, fb, tf = 0;
          field_span(T, d, f, &fa, &fb);
          for (uint32_t i = fa; i < fb; i++) tf += T->tok[i] == rk;
          if (tf) { cf[ncf] = f; cc[ncf] = tf; ncf++; }
        }
        write_field_vec(recs, ncf, cf, cc, ncf == 1 && cf[0] == T->longest_field, id_bits);
      }
    }
    write_field_vec(recs, nf, fid, cnt, only_longest, id_bits);
    for (uint32_t i = 0; i < total; i++) b_posvint(recs, dl[i]);
    const uint32_t len = (uint32_t)(recs->n - r0);
    if (psize == 2 && size_positions + len >= positions_limit) { psize = 3; pivot = r; three = 1; }
    size_positions += len;
    rec_end[n_rec++] = (uint32_t)recs->n;
    if (psize == 2) { b_u8(ptrs, size_positions & 255); b_u8(ptrs, (size_positions >> 8) & 127); }
    else { b_u8(ptrs, size_positions & 255); b_u8(ptrs, (size_positions >> 8) & 255); b_u8(ptrs, (size_positions >> 16) & 127); }
  }
  for (uint32_t i = n_rec; i > 0; i--) {
    const uint32_t s0 = i > 1 ? rec_end[i - 2] : 0u, e = rec_end[i - 1];
    b_put(bodies, recs->p + s0, e - s0);
  }
  free(rec_end); free(dl);
  b_put(bodies, ptrs->p, ptrs->n);
  uint32_t ctype;
  container_of(T, a, n, bodies, &ctype);
  *ctp_out = (ctype << 30) | (uint32_t)(base + size_positions);
  *pivot_out = pivot;
}

typedef struct { uint64_t key_hash; uint32_t key; } seg_ent;
static int seg_cmp(const void* x, const void* y) {
  const seg_ent* a = (const seg_ent*)x; const seg_ent* b = (const seg_ent*)y;
  return a->key_hash < b->key_hash ? -1 : a->key_hash > b->key_hash;
}

/* index.bin of the corpus; key_head_size 20 (no n-gram keys written) | 22 (bigram keys) | 23 (bigram + trigram keys).  The caller
 * frees *out with so_text_free_bytes.  Segment of a key = (key_hash >> 40) & mask, as oracle/ref_format.py. */
int so_text_write_index_bin(const so_text* T, uint32_t segment_number_bits, uint32_t key_head_size, uint32_t positions_limit, uint8_t** out,
                            uint64_t* out_len) {
  if (!T || !out || !out_len || (key_head_size != 20 && key_head_size != 22 && key_head_size != 23) || segment_number_bits > 16) return -1;
  const uint32_t nseg = 1u << segment_number_bits;
  const uint32_t n_levels = (uint32_t)((T->n_docs + 65535) >> 16);
  buf F = {0, 0, 0}, heads = {0, 0, 0}, bodies = {0, 0, 0}, payload = {0, 0, 0}, tbl = {0, 0, 0}, recs = {0, 0, 0}, ptrs = {0, 0, 0};
  b_le(&F, 6, 2); b_le(&F, 1, 2);
  /* per key: cursor into its postings (levels ascend) */
  uint64_t* cur = (uint64_t*)malloc((size_t)T->n_keys * 8);
  memcpy(cur, T->key_off, (size_t)T->n_keys * 8);
  /* keys by segment, once */
  uint32_t* seg_cnt = (uint32_t*)calloc((size_t)nseg + 1, 4);
  uint32_t n_use = 0;
  for (uint32_t k = 0; k < T->n_keys; k++) {
    if (T->key_off[k + 1] == T->key_off[k]) continue;
    if ((T->ncomp[k] > 1 && T->ncomp[k] > key_head_size - 20u)) continue;  /* a head without room for the component df bytes: such keys do not exist in that index */
    seg_cnt[((T->key_hash[k] >> 40) & (nseg - 1)) + 1]++;
    n_use++;
  }
  for (uint32_t s = 0; s < nseg; s++) seg_cnt[s + 1] += seg_cnt[s];
  seg_ent* se = (seg_ent*)malloc((size_t)(n_use ? n_use : 1) * sizeof(seg_ent));
  uint32_t* fill = (uint32_t*)malloc((size_t)nseg * 4);
  memcpy(fill, seg_cnt, (size_t)nseg * 4);
  for (uint32_t k = 0; k < T->n_keys; k++) {
    if (T->key_off[k + 1] == T->key_off[k] || (T->ncomp[k] > 1 && T->ncomp[k] > key_head_size - 20u)) continue;
    const uint32_t s = (uint32_t)((T->key_hash[k] >> 40) & (nseg - 1));
    se[fill[s]].key_hash = T->key_hash[k]; se[fill[s]].key = k; fill[s]++;
  }
  free(fill);
  for (uint32_t s = 0; s < nseg; s++) qsort(se + seg_cnt[s], seg_cnt[s + 1] - seg_cnt[s], sizeof(seg_ent), seg_cmp);
  uint64_t psum = 0;
  for (uint32_t level = 0; level < n_levels; level++) {
    if (level == 0) b_le(&F, T->longest_field, 2);  /* longest_field_id */
    const uint64_t d0 = (uint64_t)level << 16, d1 = T->n_docs < d0 + 65536 ? T->n_docs : d0 + 65536;
    for (uint32_t f = 0; f < T->n_fields; f++) {  /* document_length_compressed_array of every indexed field */
      const uint8_t* dlf = T->n_fields > 1 ? T->fdoclen + (size_t)f * T->n_docs : T->doclen;
      b_put(&F, dlf + d0, d1 - d0);
      for (uint64_t z = d1 - d0; z < 65536; z++) b_u8(&F, 0);
      for (uint64_t d = d0; d < d1; d++) psum += so_byte4_to_int(dlf[d]);
    }
    b_le(&F, d1, 8);
    b_le(&F, psum, 8);
    tbl.n = 0; payload.n = 0;
    for (uint32_t s = 0; s < nseg; s++) {
      heads.n = 0; bodies.n = 0;
      uint32_t nk = 0;
      for (uint32_t e = seg_cnt[s]; e < seg_cnt[s + 1]; e++) {
        const uint32_t k = se[e].key;
        const uint64_t a = cur[k];
        uint64_t b = a;
        while (b < T->key_off[k + 1] && T->p_doc[b] < d1) b++;
        if (b == a) continue;
        cur[k] = b;
        uint32_t ctp, pivot;
        if (T->n_fields > 1) encode_key_body_fields(T, k, a, b, positions_limit, &bodies, &recs, &ptrs, &ctp, &pivot);
        else encode_key_body(T, k, a, b, positions_limit, &bodies, &recs, &ptrs, &ctp, &pivot);
        b_le(&heads, T->key_hash[k], 8);
        b_le(&heads, (uint32_t)(b - a) - 1u, 2);
        b_le(&heads, 0, 4);  /* max_docid, max_p_docid: the device image derives its own bounds */
        for (uint32_t x = 0; x < key_head_size - 20u; x++)
          b_u8(&heads, x < T->ncomp[k] && T->ncomp[k] > 1 ? so_int_to_byte4(T->term_df[T->comp[3 * (size_t)k + x]]) : 0u);
        b_le(&heads, pivot, 2);
        b_le(&heads, ctp, 4);
        nk++;
      }
      b_le(&tbl, heads.n + bodies.n, 4);
      b_le(&tbl, nk, 4);
      b_put(&payload, heads.p, heads.n);
      b_put(&payload, bodies.p, bodies.n);
    }
    b_put(&F, tbl.p, tbl.n);
    b_put(&F, payload.p, payload.n);
  }
  free(cur); free(seg_cnt); free(se); free(heads.p); free(bodies.p); free(payload.p); free(tbl.p); free(recs.p); free(ptrs.p);
  *out = F.p; *out_len = F.n;
  return 0;
}
void so_text_free_bytes(uint8_t* p) { free(p); }
