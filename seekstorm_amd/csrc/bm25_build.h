// Image builder entry for shards with merged per-term lists (ss_common.h bm_merged).
#pragma once
#include "ss_common.h"

// merged_boost: the real fields' boosts; the last list of every term is built as the merged list -- its offs / docs give the
// docs, its weights come from the term's field lists.  *merged_scale receives the power of two the weights were divided by
// (chosen from the corpus: the smallest that brings the largest merged weight under the weight code's 4.0).
// SS_MERGED_RANGE: the merged weights of this corpus span more than the code's range (boosts very far apart) -- nothing was
// built, the caller builds the image without merged lists.
constexpr int SS_MERGED_RANGE = 1000;
int ssi_bm25_build_from_host_merged(ss_shard* s, const uint8_t* doclen, const uint64_t* offs, const uint32_t* docs, const uint16_t* tfs,
                                    uint64_t positions_sum, const float* merged_boost, float* merged_scale);
