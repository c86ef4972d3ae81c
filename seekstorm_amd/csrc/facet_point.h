// Point (geo) facets: the facet routines of ss_common.h with the base point the distances are measured from
// (facet.hip; nullptr = the facet's stored bits, as the declarations in ss_common.h behave).
#pragma once
#include "ss_common.h"

int ssi_facet_kth(ss_shard* s, const unsigned long long* d_bits, uint64_t n_docs, uint64_t n_matches, uint32_t offset, uint32_t type,
                  bool descending, uint64_t k, unsigned long long* d_hist, uint64_t* value_bits, uint64_t* n_better, uint64_t* n_equal,
                  const ss_facet_point* point, hipStream_t st);
int ssi_facet_values(ss_shard* s, const uint32_t* d_docs, uint32_t n, uint32_t offset, uint32_t type, unsigned long long* d_out,
                     const ss_facet_point* point, hipStream_t st);
int ssi_facet_count(ss_shard* s, const unsigned long long* d_bits, uint64_t n_docs, uint32_t offset, uint32_t type, uint32_t n_buckets,
                    const uint64_t* d_bounds, unsigned long long* d_counts, const ss_facet_point* point, hipStream_t st);
