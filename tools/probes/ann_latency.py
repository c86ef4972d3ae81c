"""ANN-mode latency on the C3 image (10M x 768): device-resident queries, HIP events around ss_vec_search_*_ann_dev.
Synthetic rows carry no cluster semantics; the cluster STRUCTURE (256 clusters of 256 rows per 65 536-row level) is what
the cost depends on.  Prints ms per call for AnnMode::All and Nprobe(n) at batch 1 and 64, f32 and i8."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes as C
import seekstorm_amd as S
from seekstorm_amd import _native as N
from oracle import oracle as O

n_rows, dim, k = int(os.environ.get("ANN_ROWS", 10_000_000)), 768, 100
dev = torch.device("cuda", 0)
lvl = 65536
n_levels = (n_rows + lvl - 1) // lvl
lc, cc = [], []
for l in range(n_levels):
    n = min(lvl, n_rows - l * lvl)
    c = [256] * (n // 256) + ([n % 256] if n % 256 else [])
    lc.append(len(c)); cc += c
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
sptr = C.c_void_p(stream.cuda_stream)
o_doc = torch.empty((64, k), dtype=torch.int32, device=dev); o_score = torch.empty((64, k), dtype=torch.float32, device=dev)
o_cnt = torch.empty((64,), dtype=torch.int32, device=dev); o_tot = torch.empty((64,), dtype=torch.int64, device=dev)
o_ncl = torch.empty((64,), dtype=torch.int32, device=dev)
qf = O.vec_gen(O.VECQ_SEED, 0, 64, dim)
res = {}
for prec in os.environ.get("ANN_PREC", "f32,i8").split(","):
    sh = S.Shard(0)
    if prec == "f32":
        sh.synth_vectors(O.VEC_SEED, n_rows, dim)
        q = torch.from_numpy(qf).to(dev)
        esz = 4
    else:
        sh.synth_vectors_i8(O.VEC_SEED, n_rows, dim)
        q = torch.from_numpy(O.quantize_i8(qf)).to(dev)
        esz = 1
    sh.set_clusters(lc, cc)
    def call(nq, mode):
        m = None if mode is None else mode._c()
        mp = None if m is None else C.addressof(m)
        if prec == "f32":
            N.check(S.lib().ss_vec_search_ann_dev(sh._h, nq, q.data_ptr(), k, N.FLT_MIN_NEG, mp, o_doc.data_ptr(), o_score.data_ptr(),
                                                  o_cnt.data_ptr(), o_tot.data_ptr(), o_ncl.data_ptr(), sptr), "dev")
        else:
            N.check(S.lib().ss_vec_search_i8_ann_dev(sh._h, nq, q.data_ptr(), None, k, N.FLT_MIN_NEG, mp, o_doc.data_ptr(),
                                                     o_score.data_ptr(), o_cnt.data_ptr(), o_tot.data_ptr(), o_ncl.data_ptr(), sptr), "dev")
    for nq in (1, 64):
        for name, mode in (("all", None), ("nprobe4", S.AnnMode.Nprobe(4)), ("nprobe16", S.AnnMode.Nprobe(16)), ("nprobe64", S.AnnMode.Nprobe(64))):
            if name not in os.environ.get("ANN_MODES", "all,nprobe4,nprobe16,nprobe64").split(","):
                continue
            for _ in range(3):
                call(nq, mode)
            torch.cuda.synchronize()
            ts = []
            for _ in range(12):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream); call(nq, mode); b.record(stream); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            res[f"{prec}_b{nq}_{name}"] = round(float(np.median(ts)), 4)
            res[f"{prec}_b{nq}_{name}_count_min_max"] = [int(o_cnt[:nq].min().item()), int(o_cnt[:nq].max().item())]
            if mode is not None:
                res[f"{prec}_b{nq}_{name}_clusters"] = int(o_ncl[0].item())
    sh.close()
print(json.dumps(res))
