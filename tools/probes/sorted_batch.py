"""result sort at the C2 size: 64 3-term OR queries sorted by (date desc) and by (category asc, price desc), k = 10 --
ss_bm25_search_sorted (pivots on the device, one call for the batch) against the host composition (a pivot per radix byte and a
filtered search per level through the older entry points, one query at a time)"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import seekstorm_amd as S
from oracle import oracle as O
import bench

n_docs = 10_000_000
th = O.term_thresholds()
sh = S.Shard(0)
sh.synth_lexical(O.LEX_SEED, n_docs, th, O.len_table())
rng = np.random.default_rng(3)
rec = np.dtype([("date", "<u4"), ("cat", "u1"), ("price", "<f4")])
v = np.zeros(n_docs, rec)
v["date"] = rng.integers(0, 1 << 31, n_docs); v["cat"] = rng.integers(0, 20, n_docs); v["price"] = rng.random(n_docs) * 1000
sh.upload_facets(v.view(np.uint8).reshape(n_docs, rec.itemsize))
tl = bench.make_c2_queries(O, 64)[0]
q = sh.make_queries(tl, S.QueryType.Union)
off = {n: rec.fields[n][1] for n in rec.names}
for name, spec in (("date desc", [(off["date"], "u32", True)]), ("cat asc, price desc", [(off["cat"], "u8", False), (off["price"], "f32", True)])):
    sh.search_lexical_sorted_batch(q, spec, 10)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        bd, bs, bc, bt = sh.search_lexical_sorted_batch(q, spec, 10)
    t_b = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    same = True
    for i in range(len(q)):
        cd, cs, ctot = sh.search_lexical_sorted_composed(q[i:i + 1], spec, 10)
        same &= ctot == int(bt[i]) and len(cd) == int(bc[i]) and np.allclose(cs, bs[i][:bc[i]], rtol=1e-6) and \
            all(np.array_equal(v[n][cd], v[n][bd[i][:bc[i]]]) for n in ("date", "cat", "price"))
    t_c = time.perf_counter() - t0
    print(f"sort by ({name}), 64 queries, mean matches {float(bt.mean()):.0f}: batched call {t_b * 1e3:.1f} ms ({64 / t_b:.0f} q/s), "
          f"host composition {t_c * 1e3:.1f} ms ({64 / t_c:.0f} q/s), same answers: {bool(same)}")
