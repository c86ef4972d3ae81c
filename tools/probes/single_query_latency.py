"""Single-query latency (device-resident query, HIP events) against the partition count P (SS_BM25_P)."""
import sys, os, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    import numpy as np, torch, ctypes as C
    import seekstorm_amd as S
    from seekstorm_amd import _native as N
    from oracle import oracle as O
    dev = torch.device("cuda", 0)
    sh = S.Shard(0)
    th = O.term_thresholds()
    sh.synth_lexical(O.LEX_SEED, 10_000_000, th, O.len_table())
    df = th.astype(np.float64) / 2.0 ** 32
    bands = [np.nonzero((df >= a) & (df < b))[0] for a, b in ((0.005, 0.02), (0.02, 0.05), (0.05, 0.15))]
    rng = np.random.default_rng(3)
    tl = [[int(rng.choice(b)) for b in bands] for _ in range(64)]
    q_np = sh.make_queries(tl, S.QueryType.Union)
    q_dev = torch.from_numpy(q_np.view(np.uint8).reshape(64, -1).copy()).to(dev)
    k = 10
    o_doc = torch.empty((64, k), dtype=torch.int32, device=dev); o_score = torch.empty((64, k), dtype=torch.float32, device=dev)
    o_cnt = torch.empty((64,), dtype=torch.int32, device=dev); o_tot = torch.empty((64,), dtype=torch.int64, device=dev)
    stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
    sptr = C.c_void_p(stream.cuda_stream)
    qsz = q_np.dtype.itemsize
    def one(i, rt):
        N.check(S.lib().ss_bm25_search_dev(sh._h, 1, q_dev.data_ptr() + i * qsz, k, rt, 2 | (3 << 8), o_doc.data_ptr(), o_score.data_ptr(),
                                          o_cnt.data_ptr(), o_tot.data_ptr(), sptr), "dev")
    res = {}
    for rt, name in ((N.RT_TOPK, "topk"), (N.RT_TOPKCOUNT, "topkcount")):
        for i in range(8):
            one(i, rt)
        torch.cuda.synchronize()
        ts = []
        for i in range(64):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream); one(i, rt); b.record(stream); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        res[name] = float(np.median(ts))
    print(json.dumps(res))
else:
    for P in ("", "128", "256"):
        env = dict(os.environ)
        if P:
            env["SS_BM25_P"] = P
        out = subprocess.run([sys.executable, __file__, "run"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        print("P =", P or "default", out)
