"""ctypes binding of libseekstorm_hip.so (the C ABI in include/seekstorm_hip.h).

There is no CPU fallback: if the HIP library is missing or no MI355X is visible, the calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SEEKSTORM_HIP_LIB") or os.path.join(_HERE, "lib", "libseekstorm_hip.so")  # override: experiment builds (tools/probes)

SS_NO_DOC = 0xFFFFFFFF
SS_OK, SS_EINVAL, SS_ENOMEM, SS_EDEVICE, SS_ENOTSUP, SS_ESTATE, SS_EPEER = 0, -1, -2, -3, -4, -5, -6
SS_MAX_QUERY_TERMS = 32
SS_MAX_PHRASE = 12
SS_PHRASE_SKIP = 0xFF
SS_MAX_K = 1024
SS_VEC_BATCH = 64
OP_INTERSECTION, OP_UNION, OP_PHRASE = 0, 1, 2
BM25_AUTO, BM25_EXHAUSTIVE, BM25_PRUNED, BM25_EXHAUSTIVE_F32 = 0, 1, 2, 3
RT_COUNT, RT_TOPK, RT_TOPKCOUNT = 0, 1, 2
MODE_LEXICAL, MODE_VECTOR, MODE_HYBRID = 0, 1, 2
SRC_LEXICAL, SRC_VECTOR, SRC_HYBRID = 0, 1, 2
SIM_DOT, SIM_EUCLIDEAN = 0, 1
FLT_MIN_NEG = -3.4028234663852886e38

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)


class Bm25Query(C.Structure):
    _fields_ = [("n_terms", C.c_uint32), ("op", C.c_uint32), ("term", C.c_uint32 * SS_MAX_QUERY_TERMS),
                ("idf", C.c_float * SS_MAX_QUERY_TERMS), ("phrase_len", C.c_uint32), ("phrase_seq", C.c_uint8 * SS_MAX_PHRASE)]


class RefBlock(C.Structure):  # ss_ref_block
    _fields_ = [("block_id", C.c_uint32), ("compression_type_pointer", C.c_uint32), ("posting_count_m1", C.c_uint16),
                ("pointer_pivot_p_docid", C.c_uint16), ("byte_array", C.c_void_p), ("byte_array_len", C.c_uint64)]


class FacetFilterC(C.Structure):  # ss_facet_filter
    _fields_ = [("offset", C.c_uint32), ("type", C.c_uint32), ("lo", C.c_uint64), ("hi", C.c_uint64), ("n_values", C.c_uint32),
                ("values", C.c_uint32 * 8), ("reserved", C.c_uint32)]


SS_MAX_FACET_FILTERS = 8
FACET_HI_INCLUSIVE, FACET_LO_EXCLUSIVE = 1, 2
FACET_IDS_EXTERN = 0xFFFFFFFF
FACET_TYPES = {"u8": 0, "u16": 1, "u32": 2, "u64": 3, "i8": 4, "i16": 5, "i32": 6, "i64": 7, "f32": 8, "f64": 9,
               "string16": 10, "string32": 11, "point": 12}
POINT_UNITS = {"sortkey": 0, "km": 1, "miles": 2}


class FacetPointC(C.Structure):  # ss_facet_point
    _fields_ = [("lat", C.c_double), ("lon", C.c_double), ("unit", C.c_uint32), ("reserved", C.c_uint32)]


class AnnModeC(C.Structure):  # ss_ann_mode
    _fields_ = [("n_probe", C.c_uint32), ("cluster_threshold_raw", C.c_float), ("field_mask", C.c_uint64), ("flags", C.c_uint32),
                ("reserved", C.c_uint32)]


SS_ANN_REPORT_OBSERVED = 1


class ResultSortC(C.Structure):  # ss_result_sort
    _fields_ = [("facet_offset", C.c_uint32), ("facet_type", C.c_uint32), ("descending", C.c_uint32), ("reserved", C.c_uint32),
                ("base_lat", C.c_double), ("base_lon", C.c_double)]


SS_MAX_SORT_FIELDS = 4


class VecLevelC(C.Structure):  # ss_vec_level
    _fields_ = [("n_rows", C.c_uint64), ("rows", C.c_void_p), ("elem_i8", C.c_uint32), ("n_clusters", C.c_uint32), ("row_doc_ids", C.c_void_p),
                ("row_scale", C.c_void_p), ("row_norm", C.c_void_p), ("row_field", C.c_void_p), ("child_count", C.c_void_p)]


BM25_QUERY_DTYPE = np.dtype([("n_terms", np.uint32), ("op", np.uint32), ("term", np.uint32, (SS_MAX_QUERY_TERMS,)),
                             ("idf", np.float32, (SS_MAX_QUERY_TERMS,)), ("phrase_len", np.uint32), ("phrase_seq", np.uint8, (SS_MAX_PHRASE,))])
assert BM25_QUERY_DTYPE.itemsize == C.sizeof(Bm25Query)

# every symbol include/seekstorm_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("ss_abi_version", C.c_int, []),
    ("ss_strerror", C.c_char_p, [C.c_int]),
    ("ss_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("ss_shard_create", C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    ("ss_shard_destroy", C.c_int, [C.c_void_p]),
    ("ss_shard_sync", C.c_int, [C.c_void_p]),
    ("ss_set_deleted", C.c_int, [C.c_void_p, u64p, C.c_uint64]),
    ("ss_bm25_upload", C.c_int, [C.c_void_p, C.c_uint64, u8p, C.c_uint32, u64p, u32p, u16p]),
    ("ss_bm25_upload_positions", C.c_int, [C.c_void_p, C.c_uint64, u8p, C.c_uint32, u64p, u32p, u16p, u16p, C.c_uint64]),
    ("ss_bm25_fields_info", C.c_int, [C.c_void_p, u32p, u32p, u32p]),
    ("ss_bm25_upload_fields", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, u8p, f32p, C.c_uint32, u64p, u32p, u8p, u16p]),
    ("ss_bm25_upload_fields_positions", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, u8p, f32p, C.c_uint32, u64p, u32p, u8p, u16p, u16p,
                                                  C.c_uint64]),
    ("ss_ref_decode_block", C.c_int, [C.c_void_p, u16p, u16p]),
    ("ss_bm25_append_level", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, u8p, C.c_uint32, u64p, u32p, u16p]),
    ("ss_bm25_append_level_fields", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, u8p, f32p, C.c_uint32, u64p, u32p, u8p, u16p]),
    ("ss_bm25_append_level_positions", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, u8p, C.c_uint32, u64p, u32p, u16p, u16p, u16p, C.c_uint64]),
    ("ss_bm25_append_sparse_level", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, u64p, u32p, u16p, u16p, u16p, C.c_uint64]),
    ("ss_bm25_incremental_info", C.c_int, [C.c_void_p, u32p, u64p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("ss_bm25_upload_ref_blocks", C.c_int, [C.c_void_p, C.c_uint64, u8p, C.c_uint32, u64p, C.c_void_p]),
    ("ss_index_bin_open", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    ("ss_index_bin_filter", C.c_int, [C.c_void_p, C.c_uint64, u32p]),
    ("ss_index_bin_decode_stats", C.c_int, [C.c_void_p, C.c_int, u64p, u64p]),
    ("ss_index_bin_decode_all", C.c_int, [C.c_void_p, u64p, u32p, u16p, C.c_uint64, u16p, u16p, C.c_uint64]),
    ("ss_index_bin_close", C.c_int, [C.c_void_p]),
    ("ss_index_bin_info", C.c_int, [C.c_void_p, u64p, u64p, u32p, u32p, u32p]),
    ("ss_index_bin_term_keys", C.c_int, [C.c_void_p, u64p]),
    ("ss_index_bin_term_ngram", C.c_int, [C.c_void_p, u8p, u8p, u32p]),
    ("ss_ref_decode_block_ngram", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, u16p, u16p]),
    ("ss_ref_decode_block_fields_ngram_positions", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, u16p, u32p, u8p, u16p, u16p, u16p, C.c_uint64, u64p]),
    ("ss_ref_decode_block_ngram_positions", C.c_int, [C.c_void_p, C.c_uint32, u16p, u16p, u16p, u16p, C.c_uint64, u64p]),
    ("ss_index_bin_term_postings", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, u32p, u16p, u64p]),
    ("ss_bm25_upload_index_bin", C.c_int, [C.c_void_p, C.c_void_p]),
    ("ss_bm25_upload_index_bin_positions", C.c_int, [C.c_void_p, C.c_void_p]),
    ("ss_ref_decode_block_positions", C.c_int, [C.c_void_p, u16p, u16p, u16p, C.c_uint64, u64p]),
    ("ss_bm25_upload_index_bin_fields", C.c_int, [C.c_void_p, C.c_void_p, f32p]),
    ("ss_ref_decode_block_fields", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, u16p, u32p, u8p, u16p]),
    ("ss_ref_decode_block_fields_positions", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, u16p, u32p, u8p, u16p, u16p, C.c_uint64, u64p]),
    ("ss_bm25_upload_index_bin_fields_positions", C.c_int, [C.c_void_p, C.c_void_p, f32p]),
    ("ss_ref_decode_block_fields_ngram", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u16p, u32p, u8p, u16p]),
    ("ss_synth_set_partition", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    ("ss_bm25_synth", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, u32p, u8p]),
    ("ss_bm25_info", C.c_int, [C.c_void_p, u64p, f32p, u32p, u64p]),
    ("ss_bm25_term_df", C.c_int, [C.c_void_p, C.c_uint32, u32p, u64p]),
    ("ss_bm25_set_strategy", C.c_int, [C.c_void_p, C.c_int]),
    ("ss_bm25_set_probe_budget", C.c_int, [C.c_void_p, C.c_uint64]),
    ("ss_bm25_term_probed", C.c_int, [C.c_void_p, C.c_uint32, u32p, u8p]),
    ("ss_bm25_search", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, u32p, f32p, u32p, u64p]),
    ("ss_bm25_search_dev", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("ss_vec_upload", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, f32p, u32p]),
    ("ss_vec_upload_vector_bin", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]),
    ("ss_vec_synth", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32]),
    ("ss_vec_info", C.c_int, [C.c_void_p, u64p, u32p]),
    ("ss_vec_read_rows", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, f32p]),
    ("ss_vec_search", C.c_int, [C.c_void_p, C.c_uint32, f32p, C.c_uint32, C.c_float, u32p, f32p, u32p, u64p]),
    ("ss_vec_search_dev", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    ("ss_vec_upload_i8", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, f32p, u32p]),
    ("ss_vec_upload_vector_bin_i8", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int]),
    ("ss_vec_synth_i8", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32]),
    ("ss_vec_read_rows_i8", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    ("ss_vec_search_i8", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, f32p, C.c_uint32, C.c_float, u32p, f32p, u32p, u64p]),
    ("ss_vec_search_i8_dev", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("ss_facet_upload", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]),
    ("ss_bm25_search_filtered", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, u32p,
                                          f32p, u32p, u64p]),
    ("ss_bm25_search_filtered_dev", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("ss_bm25_facet_count", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, u64p, u64p,
                                      u64p]),
    ("ss_vec_set_similarity", C.c_int, [C.c_void_p, C.c_int]),
    ("ss_vec_set_row_norms", C.c_int, [C.c_void_p, C.c_uint64, f32p]),
    ("ss_vec_search_i8_euclid", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, f32p, f32p, C.c_uint32, C.c_float, C.c_void_p, u32p, f32p,
                                          u32p, u64p, u32p]),
    ("ss_vec_search_i8_euclid_dev", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("ss_vec_set_clusters", C.c_int, [C.c_void_p, C.c_uint32, u32p, u32p]),
    ("ss_vec_cluster_info", C.c_int, [C.c_void_p, u32p, u32p]),
    ("ss_vec_set_fields", C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    ("ss_vec_search_ann", C.c_int, [C.c_void_p, C.c_uint32, f32p, C.c_uint32, C.c_float, C.c_void_p, u32p, f32p, u32p, u64p, u32p]),
    ("ss_vec_search_ann_dev", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("ss_vec_search_i8_ann", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, f32p, C.c_uint32, C.c_float, C.c_void_p, u32p, f32p,
                                       u32p, u64p, u32p]),
    ("ss_vec_search_i8_ann_dev", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("ss_merge_results", C.c_int, [C.c_int, u64p, f32p, C.c_uint32, u64p, f32p, C.c_uint32, C.c_uint32, C.c_uint32,
                                   u64p, f32p, u8p]),
    ("ss_topk_merge_dev", C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("ss_topk_merge_dev_packed", C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    ("ss_topk_concat_dev_packed", C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]),
    ("ss_comm_unique_id", C.c_int, [C.c_void_p]),
    ("ss_comm_create", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    ("ss_comm_create_all", C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    ("ss_comm_destroy", C.c_int, [C.c_void_p]),
    ("ss_comm_info", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("ss_topk_allgather_merge", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    ("ss_bm25_facet_kth", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                    u64p, u64p, u64p, u64p]),
    ("ss_facet_values", C.c_int, [C.c_void_p, C.c_uint32, u32p, C.c_uint32, C.c_uint32, u64p]),
    ("ss_bm25_facet_count_point", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, u64p,
                                            u64p, u64p]),
    ("ss_bm25_facet_kth_point", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                          C.c_uint64, u64p, u64p, u64p, u64p]),
    ("ss_facet_point_distances", C.c_int, [C.c_void_p, C.c_uint32, u32p, C.c_uint32, C.c_void_p, u64p]),
    ("ss_bm25_search_sharded", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    ("ss_rrf_merge_dev", C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int,
                                   C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("ss_vec_search_sharded", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    ("ss_hybrid_search_sharded", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_float, C.c_uint32,
                                           C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("ss_comm_profile", C.c_int, [C.c_void_p, C.c_int]),
    ("ss_comm_profile_read", C.c_int, [C.c_void_p, u64p, C.POINTER(C.c_double), C.c_int]),
    ("ss_index_bin_tier", C.c_int, [C.c_void_p, C.c_uint64, u32p]),
    ("ss_bm25_search_sorted", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                        u32p, f32p, u32p, u64p]),
    ("ss_vec_append_rows", C.c_int, [C.c_void_p, C.c_void_p]),
    ("ss_vec_reserve_rows", C.c_int, [C.c_void_p, C.c_uint64]),
    ("ss_bm25_append_sparse", C.c_int, [C.c_void_p, C.c_uint32, u64p, u32p, u16p, u32p]),
    ("ss_bm25_append_sparse_fields", C.c_int, [C.c_void_p, C.c_uint32, u64p, u32p, u8p, u16p, u32p]),
    ("ss_bm25_append_sparse_positions", C.c_int, [C.c_void_p, C.c_uint32, u64p, u32p, u16p, u16p, C.c_uint64, u16p, u32p]),
    ("ss_bm25_append_sparse_fields_positions", C.c_int, [C.c_void_p, C.c_uint32, u64p, u32p, u8p, u16p, u16p, C.c_uint64, u16p, u32p]),
    ("ss_bm25_sparse_info", C.c_int, [C.c_void_p, u32p, u64p, u64p]),
    ("ss_shard_set_coalescing", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
    ("ss_shard_coalescing_stats", C.c_int, [C.c_void_p, u64p, u64p, u64p, u64p]),
    ("ss_bm25_path_stats", C.c_int, [C.c_void_p, u64p]),
    ("ss_bm25_shape_stats", C.c_int, [C.c_void_p, u64p]),
    ("ss_profile_enable", C.c_int, [C.c_void_p, C.c_int]),
    ("ss_profile_read", C.c_int, [C.c_void_p, C.c_int, u64p, C.POINTER(C.c_double), C.c_int]),
]

_lib = None


class SeekStormHipError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib().ss_strerror(code).decode() if _lib is not None else "?"
        super().__init__(f"{where}: {msg} (code {code})")


def lib():
    """Load the HIP library.  Raises if it has not been built (python -m seekstorm_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python seekstorm_amd/build.py` (hipcc, gfx950). "
                               "There is no CPU fallback.")
        # One HIP runtime per process: torch bundles its own libamdhip64 (requested by file name, SONAME
        # libamdhip64.so.7); ours asks for the SONAME.  Loaded first, torch's copy satisfies both; loaded second it
        # becomes a second runtime that finds no device ("No HIP GPUs are available").  torch is plumbing here
        # (buffers, streams, torch.distributed) and optional: without it the ROCm copy is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code, where):
    if code < 0:
        raise SeekStormHipError(code, where)
    return code


def ptr(a, t):
    return None if a is None else a.ctypes.data_as(t)
