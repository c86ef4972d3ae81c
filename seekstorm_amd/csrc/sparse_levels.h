// The sparse tier of an image that grows level by level (ss_bm25_append_level + ss_bm25_append_sparse_level; commit.rs:142-148).
// A commit changes the shard's average document length, and with it every BM25 weight (commit.rs recomputes bm25_component_cache): the
// tier therefore keeps the tf of every sparse posting beside its (code | doc) word and re-codes the postings on the device after a
// commit.  That state -- the tf array and the per-list starts of the positions pool -- lives beside the shard, keyed by it.
#pragma once
#include "ss_common.h"

bool ssi_bm25_sparse_levels_has(const ss_shard* s);
void ssi_bm25_sparse_levels_drop(const ss_shard* s);
// every sparse posting's code again from its tf, the image's d_doclen and d_comp (after a commit moved the average length)
int ssi_bm25_sparse_levels_recode(ss_shard* s, hipStream_t st);
// the postings of the rare terms in `level` (the level the dense image committed last): list i continues sparse list i (ascending
// docs, all behind the list's last), lists past the tier's current count are new terms; a level the tier has seen already is
// replaced.  Caller holds s->mu, the device is idle.
int ssi_bm25_append_sparse_level(ss_shard* s, uint32_t level, uint32_t n_lists, const uint64_t* offs, const uint32_t* docs, const uint16_t* tfs,
                                 const uint16_t* npos, const uint16_t* positions, uint64_t n_positions);
