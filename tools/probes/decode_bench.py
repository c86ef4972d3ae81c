"""the index.bin walker and decoder alone, on the host (no GPU needed): python tools/probes/decode_bench.py [n_docs] [n_fields]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from seekstorm_amd import _native as N
from oracle import textindex as TI
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.perf_counter()
T = TI.TextCorpus(11, n, n, n_frequent=64, mean_len=100.0, topic_share=0.35, n_fields=nf, longest_field=1 if nf > 1 else 0)
data = T.write_index_bin(key_head_size=23)
print("corpus + file %.1f s, %d bytes, %d postings" % (time.perf_counter() - t0, len(data), T.n_postings), flush=True)
buf = np.frombuffer(data, np.uint8)
L = N.lib()
for rep in range(3):
    ix = C.c_void_p()
    t0 = time.perf_counter()
    N.check(L.ss_index_bin_open(buf.ctypes.data, len(buf), nf, 23, 11, C.byref(ix)), "open")
    t1 = time.perf_counter()
    nd = C.c_uint32()
    N.check(L.ss_index_bin_tier(ix, 2000, C.byref(nd)), "tier")
    t2 = time.perf_counter()
    a, b = C.c_uint64(), C.c_uint64()
    N.check(L.ss_index_bin_decode_stats(ix, 1, C.byref(a), C.byref(b)), "decode")
    t3 = time.perf_counter()
    print("open %.3f s, tier %.3f s, decode (with positions) %.3f s: %d postings, %d positions -> %.0f M postings/s decode" %
          (t1 - t0, t2 - t1, t3 - t2, a.value, b.value, a.value / (t3 - t2) / 1e6), flush=True)
    L.ss_index_bin_close(ix)
