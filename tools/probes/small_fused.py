"""one-launch path of small host-pointer batches (bm25_small.hip) against the staged pipeline: end-to-end latency per call, answers compared
    python tools/probes/small_fused.py            (runs itself twice: SS_BM25_SMALL=1 / 0, and a PB sweep)"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)


def child():
    import seekstorm_amd as S
    from oracle import oracle as O
    import bench
    docs = int(os.environ.get("DOCS", 10_000_000))
    tl, th = bench.make_c2_queries(O, 1000)
    sh = S.Shard(0)
    sh.synth_lexical(O.LEX_SEED, docs, th, O.len_table())
    out = {}
    rng = np.random.default_rng(5)
    ba, bb = bench.band_terms(th, 0.01, 0.05), bench.band_terms(th, 0.05, 0.20)
    and_tl = [[int(rng.choice(ba)), int(rng.choice(bb))] for _ in range(64)]
    for name, lists, qt, rt in (("or3_topk", tl, S.QueryType.Union, S.ResultType.Topk), ("or3_topkcount", tl, S.QueryType.Union, S.ResultType.TopkCount),
                                ("and2_topkcount", and_tl, S.QueryType.Intersection, S.ResultType.TopkCount)):
        for nq in [int(x) for x in os.environ.get('NQS', '1,8,32,64').split(',')]:
            q = sh.make_queries(lists[:nq], qt)
            for _ in range(30):
                r = sh.search_lexical_batch(q, 10, rt, reference_shortcuts=False)
            n = 400
            lat = []
            for _ in range(n):
                t0 = time.perf_counter()
                r = sh.search_lexical_batch(q, 10, rt, reference_shortcuts=False)
                lat.append((time.perf_counter() - t0) * 1e6)
            lat = np.sort(lat)
            print("%-15s nq=%-3d p50 %7.1f us  p99 %7.1f us   %9.0f q/s" % (name, nq, lat[n // 2], lat[int(n * 0.99)], nq / (np.mean(lat) * 1e-6)), flush=True)
            out[f"{name}_{nq}"] = [np.asarray(x) for x in r]
    np.savez(os.environ["OUT"], **{f"{k}_{i}": v for k, r in out.items() for i, v in enumerate(r)})
    sh.close()


if os.environ.get("CHILD"):
    child()
    sys.exit(0)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
res = {}
LEGS = [("staged", {"SS_BM25_SMALL": "0"}), ("fused", {"SS_BM25_SMALL": "1"})] + [(f"fused_pb{x}", {"SS_BM25_SMALL_PB": str(x)}) for x in os.environ.get("PBS", "4,8,16,48,64").split(",") if x]
for tag, env in LEGS:
    out = os.path.join(ROOT, "gpurun_out", f"small_fused_{tag}.npz")
    print("====", tag, env, flush=True)
    e = dict(os.environ, CHILD="1", OUT=out, **env)
    rc = subprocess.call([sys.executable, os.path.abspath(__file__)], env=e)
    print("rc", rc, flush=True)
    if rc == 0:
        res[tag] = np.load(out)
if "staged" in res:
    for tag in res:
        if tag == "staged":
            continue
        bad = [k for k in res["staged"].files if not np.array_equal(res["staged"][k], res[tag][k])]
        print(tag, "vs staged:", "IDENTICAL" if not bad else f"DIFFER in {bad[:8]}")
