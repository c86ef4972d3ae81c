"""Where does the per-launch constant of the i8 scan go?  Compares a normal pass with one whose threshold no row passes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import seekstorm_amd as S
from oracle import oracle as O

sh = S.Shard(0)
sh.synth_vectors_i8(O.VEC_SEED, 10_000_000, 768)
q = O.quantize_i8(O.vec_gen(O.VECQ_SEED, 0, 64, 768))
for name, thr in (("normal", None), ("no candidates", 1e9)):
    sh.search_vector_batch_i8(q, 100, similarity_threshold_raw=thr)
    sh.profile(True)
    sh.profile_read(1, reset=True)
    for _ in range(5):
        sh.search_vector_batch_i8(q, 100, similarity_threshold_raw=thr)
    n, ms = sh.profile_read(1, reset=True)
    sh.profile(False)
    print(name, "passes", n, "ms per pass", ms / max(n, 1))
