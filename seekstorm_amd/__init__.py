"""seekstorm_amd -- MI355X-native implementation of SeekStorm's query hot path.

BM25 posting-list union/intersection with exact top-k, dense-vector brute-force cosine/dot top-k, RRF hybrid fusion
and cross-shard top-k merge, as hand-written HIP kernels for gfx950 behind a C ABI (include/seekstorm_hip.h).
This package is the host-side mirror of the reference's search interface on top of that ABI.
There is no CPU fallback: without the built HIP library and a GPU every search call raises.
"""
from ._native import LIB_PATH, SeekStormHipError, lib  # noqa: F401
from .search import (AnnMode, Index, IndexBin, QueryType, Result, ResultObject, ResultSource, ResultType, SearchMode, Shard,  # noqa: F401
                     idf_f32, merge_results, normalize_f32, rrf_merge_device, quantize_f32_to_i8, threshold_raw, turboquant_dim, turboquant_f32_to_i8)

__all__ = ["AnnMode", "Index", "IndexBin", "Shard", "QueryType", "ResultType", "SearchMode", "ResultSource", "Result", "ResultObject",
           "merge_results", "rrf_merge_device", "normalize_f32", "quantize_f32_to_i8", "turboquant_dim", "turboquant_f32_to_i8", "idf_f32", "threshold_raw", "lib", "LIB_PATH", "SeekStormHipError"]
