"""the concurrent-callers legs of bench.py alone (T host threads, ONE query per call through the C++ mirror's Index::search) on the
C2 / C3 images:  python tools/probes/concurrent_bench.py [seconds]   (SS_COALESCE_LINGER=0 switches the linger off)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import seekstorm_amd as S
from seekstorm_amd import _native as N
from oracle import oracle as O
import bench
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
docs, rows, dim = int(os.environ.get("DOCS", 10_000_000)), int(os.environ.get("ROWS", 10_000_000)), 768
sh = S.Shard(0)
tl, th = bench.make_c2_queries(O, 1000)
sh.synth_lexical(O.LEX_SEED, docs, th, O.len_table())
sh.synth_vectors(O.VEC_SEED, rows if not os.environ.get('LEX_ONLY') else 1024, dim)
qv = O.vec_gen(O.VECQ_SEED, 0, 64, dim)
_libdir = os.path.dirname(os.environ["SEEKSTORM_HIP_LIB"]) if os.environ.get("SEEKSTORM_HIP_LIB") else os.path.join(ROOT, "seekstorm_amd", "lib")  # (an experiment build: its own host library beside it)
HL = C.CDLL(os.path.join(_libdir, "libseekstorm_host.so"))
HL.ssh_index_adopt.restype = C.c_void_p
HL.ssh_index_adopt.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
HL.ssh_bench_concurrent.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_double, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.POINTER(C.c_double)]
ix = HL.ssh_index_adopt(1, (C.c_void_p * 1)(sh._h), (C.c_int * 1)(0))
flat = np.array([t for q in tl for t in q], np.uint32)
toff = np.zeros(len(tl) + 1, np.uint32); toff[1:] = np.cumsum([len(q) for q in tl])
LEGS = (("lexical", N.MODE_LEXICAL, 1000, 10), ("vector", N.MODE_VECTOR, 64, 100), ("hybrid", N.MODE_HYBRID, 64, 100))
if os.environ.get("LEX_ONLY"):
    LEGS = LEGS[:1]
ONLY = os.environ.get("ONLY")  # e.g. ONLY=hybrid:64,256 -- one leg (the SS_CO_TRACE summaries are per process)
if ONLY:
    LEGS = tuple(l for l in LEGS if l[0] == ONLY.split(":")[0])
def cpu_stat():
    """(cgroup: periods throttled, us throttled, us of CPU used) -- a process over its CPU quota is stopped for the rest of the 100 ms period"""
    d = {}
    for f in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for ln in open(f):
                k_, v_ = ln.split()
                d[k_] = int(v_)
            break
        except OSError:
            pass
    return d.get("nr_throttled", 0), d.get("throttled_usec", d.get("throttled_time", 0) // 1000), d.get("usage_usec", 0)


for name, mode, nq, length in LEGS:
    for T in (tuple(int(x) for x in os.environ["CB_T"].split(",")) if os.environ.get("CB_T") else (8, 64, 256) if os.environ.get("LEX_ONLY") else tuple(int(x) for x in ONLY.split(":")[1].split(",")) if ONLY else (1, 8, 64, 256, 1024)):
        out = (C.c_double * 5)()
        s0 = sh.coalescing_stats()
        c0, t0_ = cpu_stat(), os.times()
        N.check(HL.ssh_bench_concurrent(ix, mode, T, secs, nq, flat.ctypes.data, toff.ctypes.data, qv.ctypes.data, int(S.QueryType.Union), length,
                                        N.RT_TOPK, out), "bench")
        s1 = sh.coalescing_stats()
        lb, lq, vb, vq = (s1[i] - s0[i] for i in range(4))
        print("%-8s T=%-4d %9.0f q/s  p50 %8.1f us  p99 %9.1f us  errors %d  lexical batch %s  vector batch %s" % (
            name, T, out[0] / out[1], out[2], out[3], int(out[4]), "%.1f" % (lq / lb) if lb else "-", "%.1f" % (vq / vb) if vb else "-"), flush=True)
        if os.environ.get("SSH_BENCH_HIST"):
            c1, t1_ = cpu_stat(), os.times()
            print("         cpu: %.1f cores busy (user %.2f s + sys %.2f s over %.2f s); cgroup throttled %d periods, %.1f ms" % (
                ((t1_.user - t0_.user) + (t1_.system - t0_.system)) / out[1], t1_.user - t0_.user, t1_.system - t0_.system, out[1], c1[0] - c0[0], (c1[1] - c0[1]) / 1e3), flush=True)
sh.close()
