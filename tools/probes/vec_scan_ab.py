"""A/B of the f32 vector scan's two kernels on ONE image (C3 shape, VS_ROWS rows x 768): the barrier-per-chunk kernel
(SS_VEC_SCAN_Q=0) against the waves-on-their-own kernel at several occupancies (SS_VQ_PAD / SS_VQ_WGS, read per call).
Device-resident queries, HIP events around ss_vec_search_ann_dev (mode None = AnnMode::All); answers compared between the variants.
Needs a measurement build of the library (the variants are not in the product):
    SS_OUT_DIR=$PWD/seekstorm_amd/lib_exp1 SS_HIPCC_FLAGS="-DVS_VARIANTS=1" python -c "from seekstorm_amd import build as B; B.build()"
    SEEKSTORM_HIP_LIB=$PWD/seekstorm_amd/lib_exp1/libseekstorm_hip.so python tools/probes/vec_scan_ab.py
(-DVS_PROF=1 in addition: two waves of the big launch print their cycle counts and the shader clock they ran at.)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes as C
import seekstorm_amd as S
from seekstorm_amd import _native as N
from oracle import oracle as O

n_rows, dim, k = int(os.environ.get("VS_ROWS", 5_000_000)), 768, 100
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
sptr = C.c_void_p(stream.cuda_stream)
o_doc = torch.empty((64, k), dtype=torch.int32, device=dev); o_score = torch.empty((64, k), dtype=torch.float32, device=dev)
o_cnt = torch.empty((64,), dtype=torch.int32, device=dev); o_tot = torch.empty((64,), dtype=torch.int64, device=dev)
o_ncl = torch.empty((64,), dtype=torch.int32, device=dev)
q = torch.from_numpy(O.vec_gen(O.VECQ_SEED, 0, 64, dim)).to(dev)
sh = S.Shard(0)
sh.synth_vectors(O.VEC_SEED, n_rows, dim)
def call(nq):
    N.check(S.lib().ss_vec_search_ann_dev(sh._h, nq, q.data_ptr(), k, N.FLT_MIN_NEG, None, o_doc.data_ptr(), o_score.data_ptr(),
                                          o_cnt.data_ptr(), o_tot.data_ptr(), o_ncl.data_ptr(), sptr), "dev")
variants = [("barrier", {"SS_VEC_SCAN_Q": "0"}), ("pipe", {"SS_VEC_SCAN_Q": "2"})]
for spec in os.environ.get("VS_VARIANTS", "q1:81920:1").split(","):
    name, pad, wgs = spec.split(":")
    variants.append((name, {"SS_VEC_SCAN_Q": "1", "SS_VQ_PAD": pad, "SS_VQ_WGS": wgs}))
ref = {}
for nq in (64, 1):
    for name, env in variants:
        for kx in ("SS_VEC_SCAN_Q", "SS_VQ_PAD", "SS_VQ_WGS"):
            os.environ.pop(kx, None)
        os.environ.update(env)
        for _ in range(2):
            call(nq)
        torch.cuda.synchronize()
        ts = []
        for _ in range(8):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream); call(nq); b.record(stream); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ms = float(np.median(ts))
        # the same calls enqueued back to back (no host synchronisation in between): the shader clock ramps up over ~25 ms of
        # continuous load (tools/probes/mfma_peak.hip), a pass with an idle device before it runs at ~2.0 GHz
        nb2b = int(os.environ.get("VS_B2B", 24))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(nb2b):
            call(nq)
        b.record(stream); torch.cuda.synchronize()
        ms_b2b = a.elapsed_time(b) / nb2b
        tf = 2.0 * dim * n_rows * 64 / (ms * 1e-3) / 1e12 if nq == 64 else 0.0
        got = (o_doc[:nq].cpu().numpy().copy(), o_score[:nq].cpu().numpy().copy())
        same = "-"
        if nq in ref:
            same = "IDENTICAL" if np.array_equal(got[0], ref[nq][0]) and np.array_equal(got[1], ref[nq][1]) else "DIFFERENT"
        else:
            ref[nq] = got
        tf2 = 2.0 * dim * n_rows * 64 / (ms_b2b * 1e-3) / 1e12 if nq == 64 else 0.0
        print(f"nq={nq:<3} {name:<8} {ms:8.3f} ms  {tf:6.1f} TFLOP/s  frac {tf / 157.3:.3f}  GB/s {n_rows * dim * 4 / ms / 1e6:7.0f}  | back to back x{nb2b}: {ms_b2b:8.3f} ms  {tf2:6.1f} TFLOP/s  frac {tf2 / 157.3:.3f}  GB/s {n_rows * dim * 4 / ms_b2b / 1e6:7.0f}  vs barrier: {same}", flush=True)
sh.close()
