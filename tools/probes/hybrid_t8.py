"""hybrid callers at T = 8 / 64 (one query per call through the C++ mirror): q/s and batch sizes"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import seekstorm_amd as S
from seekstorm_amd import _native as N
from oracle import oracle as O
import bench
sh = S.Shard(0)
tl, th = bench.make_c2_queries(O, 1000)
sh.synth_lexical(O.LEX_SEED, 10_000_000, th, O.len_table())
sh.synth_vectors(O.VEC_SEED, 10_000_000, 768)
qv = O.vec_gen(O.VECQ_SEED, 0, 64, 768)
HL = C.CDLL(os.path.join(ROOT, "seekstorm_amd", "lib", "libseekstorm_host.so"))
HL.ssh_index_adopt.restype = C.c_void_p
HL.ssh_index_adopt.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
HL.ssh_bench_concurrent.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_double, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.POINTER(C.c_double)]
ix = HL.ssh_index_adopt(1, (C.c_void_p * 1)(sh._h), (C.c_int * 1)(0))
flat = np.array([t for q in tl for t in q], np.uint32)
toff = np.zeros(len(tl) + 1, np.uint32); toff[1:] = np.cumsum([len(q) for q in tl])
for T in (8, 64):
    out = (C.c_double * 5)()
    s0 = sh.coalescing_stats()
    N.check(HL.ssh_bench_concurrent(ix, N.MODE_HYBRID, T, 1.5, 64, flat.ctypes.data, toff.ctypes.data, qv.ctypes.data, int(S.QueryType.Union), 100, N.RT_TOPK, out), "bench")
    s1 = sh.coalescing_stats()
    lb, lq, vb, vq = (s1[i] - s0[i] for i in range(4))
    print("hybrid T=%-3d %8.0f q/s  p50 %8.1f us  p99 %8.1f us  lexical batch %.1f  vector batch %.1f" % (T, out[0] / out[1], out[2], out[3], lq / max(lb, 1), vq / max(vb, 1)), flush=True)
sh.close()
