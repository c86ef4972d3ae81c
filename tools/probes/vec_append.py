"""ss_vec_append_rows at the C3 shape: a 10 M x 768 f32 image (device-generated), then levels of 65 536 records appended;
the first append grows the image by half (one device-to-device copy), the following ones write in place."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import seekstorm_amd as S
from seekstorm_amd import _native as N

n0, dim, lvl = 9_900_000, 768, 65536
sh = S.Shard(0)
N.check(N.lib().ss_vec_synth(sh._h, 77, n0, dim), "ss_vec_synth")
sh.vector_count, sh.dim = n0, dim
rng = np.random.default_rng(1)
rows = rng.standard_normal((lvl, dim)).astype(np.float32)
rows /= np.linalg.norm(rows, axis=1, keepdims=True)
qs = rows[:4].copy()
for i in range(4):
    t0 = time.perf_counter()
    sh.append_vector_rows(rows)
    dt = time.perf_counter() - t0
    d, s, c, t = sh.search_vector_batch(qs, 3)
    # the appended rows are found: query i is row i of every appended level, similarity 1
    ok = all(abs(float(s[j, 0]) - 1.0) < 1e-5 and int(d[j, 0]) >= n0 for j in range(4))
    print(f"append {i}: {dt * 1e3:.1f} ms for {lvl} rows x {dim} f32 ({lvl * dim * 4 / 1e6:.0f} MB from pageable host memory); rows now {sh.vector_count}; found: {ok}")
