#!/usr/bin/env python3
"""bench.py -- SeekStorm query hot path on MI355X.

Primary workload (BASELINE.json configs[1], "C2"): 10 M synthetic docs, 3-term OR, BM25 top-10, one shard per GPU.
One C-ABI call = one batch of 1000 resolved queries through ss_bm25_search -- HOST pointers in, answers on the HOST when the call
returns (SURVEY 8d's timing protocol: submit -> results on host; round 6, VERDICT r5 "next" 6; N > 1: ss_bm25_search_sharded, which ends
in the all-gather and the merged answers on every rank's host); one "step" = --calls-per-step (default 200) such calls, so that the timed
region (exactly --steps steps) covers >= 200 batches and >= 2 s.  The device-resident rate (ss_bm25_search_dev: queries and answers in
HBM, what rounds 1-5 headlined) is the secondary key `device_resident`.  Secondary legs in the same JSON line: the exhaustive strategy (the kernel that
streams SURVEY 8d's algorithmic bytes), TopkCount, C3 (10 M x 768 f32 cosine top-100, batch 64) + i8, C4 hybrid, ANN,
host-pointer end-to-end rates, latencies (>= 1000 samples), concurrent single-query callers through the C++ mirror (T = 64 / 256
threads), the multi-shard entry points (one all-gather per call, allgather_us), full-size parity against the oracle on ALL queries
of the batches, and the CPU baseline
(the oracle's reference-structured dispatch, union_docid_3, timed on the host cores).

Multi-GPU (SURVEY 8e, weak scaling): one process per GPU; rank r holds shard r of ONE generator stream of
docs_per_shard x N docs (doc g -> shard g % N, index.rs:5284); every rank answers the same batch on its shard, one RCCL
all-gather of the per-shard top-k (behind the C ABI: ss_topk_allgather_merge), identical merge on every rank.
`value` = merged answers per second over the N-times larger corpus.

  python bench.py --gpus N --steps K --warmup W      (N > 1 without a launcher: re-executes itself under torchrun)
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: FP32 matrix peak


KERNEL_SOURCES = ("bm25.hip", "bm25_dev.h", "bm25_fast.hip", "bm25_probe.hip", "bm25_probe_body.h", "bm25_scan16.hip", "bm25_small.hip", "ss_common.h",
                  "vec8_scan.hip", "vec_scan.hip")


def kernel_source_hash():
    """hash of the kernel sources: profiles/pmc_traffic.json is only used when it was collected on THESE kernels"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "seekstorm_amd", "csrc")
    for f in KERNEL_SOURCES:  # the files that define the measured kernels (not the loaders, the ABI layer, the generators)
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


_PMC = None


def pmc_traffic(kind, key="hbm_bytes_per_launch"):
    """HBM bytes per launch from the committed PMC run (profiles/pmc_traffic.json: tools/collect_pmc.sh + pmc_summary.py,
    separate rocprofv3 --pmc passes of this command).  None when the file is absent or was collected on other kernel
    sources (its `kernel_source_hash` must equal kernel_source_hash())."""
    global _PMC
    if _PMC is None:
        try:
            _PMC = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if _PMC.get("kernel_source_hash") != kernel_source_hash():
                _PMC = {"stale": True}
        except Exception:
            _PMC = {}
    try:
        return _PMC[kind][key]
    except Exception:
        return None


def pct(a, p):
    a = sorted(a)
    return a[min(len(a) - 1, int(round(p / 100.0 * (len(a) - 1))))] if a else None


def band_terms(th, lo, hi):
    frac = th.astype(np.float64) / 2.0 ** 32
    return np.nonzero((frac >= lo) & (frac < hi))[0]


def make_c2_queries(O, n_queries, seed=1234):
    """3 distinct terms, one from each df band [0.5-2%], [2-5%], [5-15%]."""
    th = O.term_thresholds()
    bands = [band_terms(th, 0.005, 0.02), band_terms(th, 0.02, 0.05), band_terms(th, 0.05, 0.15)]
    rng = np.random.default_rng(seed)
    return [[int(rng.choice(b)) for b in bands] for _ in range(n_queries)], th


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pick(d, keys):
    return {k_: d[k_] for k_ in keys if isinstance(d, dict) and d.get(k_) is not None}


def _r(x, nd=4):
    """shorten floats for the compact line (the side file keeps full precision)"""
    if isinstance(x, float):
        return float(f"{x:.{nd + 2}g}")
    if isinstance(x, dict):
        return {k_: _r(v_, nd) for k_, v_ in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v_, nd) for v_ in x]
    return x


def compact_line(line):
    """The ONE stdout line the driver parses (<= 4 KB).  Everything else goes to bench_details.json.

    roofline = SURVEY 8(d)'s own fraction -- algorithmic bytes / live launch time -- of the kernel that STREAMS those bytes
    (bm25_scan16_kernel, the exhaustive strategy on the same batches); the pruned kernel the headline `value` is timed on answers
    without reading most of them, so its figure is counter traffic / time and is named traffic_frac (roofline.pruned)."""
    is_bm = "exhaustive" in line
    out = _pick(line, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"))
    out["vs_baseline"] = None
    out["dtype"] = line["dtype"]
    out["data"] = "synthetic"
    out["config"] = _pick(line["config"], ("workload", "docs_per_shard", "queries_per_call", "calls_per_step", "k", "result_type", "rows_per_shard", "dim", "batch"))
    rf = line.get("roofline") or {}
    if is_bm:
        ex = (line["exhaustive"].get("roofline") or {})
        roof = _pick(ex, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms", "launches"))
        roof["kernel"] = "bm25_scan16_kernel<3,1> (exhaustive strategy, same batches: streams SURVEY 8d's bytes)"
        roof["exhaustive_value_qps"] = line["exhaustive"].get("value")
        roof["pruned"] = {"kernel": "bm25_probe_kernel<3,1> (the kernel `value` is timed on; prunes, so priced on counter traffic)",
                          "traffic_frac": rf.get("frac_counter"), "traffic": rf.get("traffic"), "avg_launch_ms": rf.get("avg_launch_ms"),
                          "launches": rf.get("launches")}
        # (round 6, VERDICT r5 "next" 5: the forced-EXHAUSTIVE fractions of 2-term ANDs, NOT + tombstones and the clustered corpus are no
        # longer in the line -- AUTO sends those shapes to the pruned kernel, see `off_uniform` below; bench_details.json keeps them.  What
        # stays is what only the scans can do: unions of more than 4 lists, and count mode without a probe index on the f32 tile)
        if rf.get("fallback_f32_frac") is not None:
            roof["fallback_f32_frac"] = rf["fallback_f32_frac"]
        u16 = ((line.get("union16") or {}).get("roofline") or {}).get("frac")
        if u16 is not None:
            roof["union16_frac"] = u16
    else:
        roof = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "mfma_util_pmc", "algorithmic_flops_per_launch",
                          "algorithmic_bytes_per_launch", "avg_launch_ms", "launches"))
    out["roofline"] = roof
    if is_bm:
        # the shapes off the uniform corpus, as a caller gets them (AUTO) beside the forced exhaustive scan, queries/s
        ou = {}
        it_, ex_, cl_ = line.get("intersection") or {}, line.get("exhaustive_not_tombstones") or {}, line.get("clustered") or {}
        if it_.get("value"):
            ou["and2_topkcount"] = {"auto": it_["value"], "exhaustive": (it_.get("exhaustive") or {}).get("value")}
        if (ex_.get("auto") or {}).get("value"):
            ou["or3_not_tombstones_topkcount"] = {"auto": ex_["auto"]["value"], "exhaustive": ex_.get("value")}
        if (cl_.get("auto_topk") or {}).get("value"):
            ou["clustered_or3_topk"] = {"auto": cl_["auto_topk"]["value"], "exhaustive": (cl_.get("exhaustive_topk") or {}).get("value")}
            ou["clustered_or3_topkcount"] = {"auto": (cl_.get("auto") or {}).get("value"), "exhaustive": (cl_.get("exhaustive") or {}).get("value")}
        if ou:
            ou["note"] = "AUTO = the pruned kernel (+ counts from the bit records); exhaustive = bm25_scan16 forced"
            out["off_uniform"] = ou
    cb = line.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "threads", "kind", "algorithm", "shards"))
        out["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
    lat = line.get("latency_ms") or {}
    out["latency_ms"] = _pick(lat, ("batch_p50", "batch_p99", "single_query_p50", "single_query_p99", "batch64_p50", "batch64_p99"))
    e2e = line.get("end_to_end") or {}
    if e2e:
        out["end_to_end"] = _pick(e2e, ("batch_ms_p50", "batch_ms_p99", "single_query_ms_p50", "single_query_ms_p99", "batch64_ms_p50"))
    if line.get("device_resident"):
        out["device_resident"] = _pick(line["device_resident"], ("value", "unit", "ms_per_call", "entry_point"))
    v = line.get("vector")
    if v:
        vr = v.get("roofline") or {}
        out["vector"] = {"metric": "queries/sec (C3: cosine top-100, 10M x 768 f32, batch 64)", "value": v.get("value"), "ms_per_call": v.get("ms_per_call"),
                         "roofline": _pick(vr, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "mfma_util_pmc",
                                                "algorithmic_flops_per_launch", "avg_launch_ms", "launches")),
                         "cpu_baseline": _pick(v.get("cpu_baseline") or {}, ("value", "unit", "cores", "kind"))}
        hy = v.get("hybrid") or {}
        if hy:
            out["hybrid"] = _pick(hy, ("value", "unit", "ms_per_call", "batch_ms_p50"))
    cc = line.get("concurrent_callers")
    if cc:
        out["concurrent_callers"] = {n_: _pick(l_, ("value", "threads", "latency_us_p50", "latency_us_p99")) for n_, l_ in cc.items() if isinstance(l_, dict)}
    if line.get("scale_check"):
        sc = line["scale_check"]
        out["scale_check"] = {"agree": sc.get("agree"), "ranks": [_pick(e, ("rank", "ranks_seen", "allgather_us", "merged_checksum_dev")) for e in sc.get("ranks", [])]}
    if line.get("parity_full_size"):
        out["parity_full_size"] = {n_: (p_.get("queries") if isinstance(p_, dict) else p_) for n_, p_ in line["parity_full_size"].items()}
    out["details"] = "bench_details.json (every secondary leg; also on stderr)"
    out = _r(out)
    # hard bound: the driver keeps a bounded tail of stdout and must find ONE parseable line in it
    for drop in ("off_uniform", "parity_full_size", "concurrent_callers", "hybrid", "end_to_end", "latency_ms", "device_resident"):
        if len(json.dumps(out)) <= 3900:
            break
        out.pop(drop, None)
    assert len(json.dumps(out)) <= 4096, len(json.dumps(out))
    return out


_REAL_STDOUT = None


def claim_stdout():
    """stdout carries ONE line, the result: everything else any library writes to file descriptor 1 (RCCL prints a five-line version banner
    through C stdio when its first communicator is made) goes to stderr from here on"""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def print_line(text):
    sys.stdout.flush()
    try:
        C.CDLL(None).fflush(None)  # (C stdio's buffer for fd 1 -- which is stderr by now -- before the line, not behind it)
    except Exception:
        pass
    if _REAL_STDOUT is None:
        print(text, flush=True)
    else:
        os.write(_REAL_STDOUT, (text + "\n").encode())


def emit(line, world):
    """details -> bench_details.json (+ stderr); the compact line -> the LAST line of stdout"""
    text = json.dumps(line, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, "bench_details.json" if world == 1 else f"bench_details_n{world}.json"), "w") as f:
                    f.write(text)
        except OSError:
            pass
    print("BENCH_DETAILS " + json.dumps(line), file=sys.stderr, flush=True)
    print_line(json.dumps(compact_line(line)))


def one_process_main(args):
    """bench.py --one-process S: C2 through the C++ mirror's Index::search_lexical_batch over S shards of ONE process (VERDICT r5 "next" 9).
    Shard i = the docs g % S == i of one generator stream of S x --docs docs on GPU i; a call = 1000 queries resolved per shard (shard-local
    idf), one task (thread) per shard, each ending in ss_bm25_search_sharded: search, ONE all-gather, merge on the device, answers on the
    host.  Same timing protocol and line as the rank-per-GPU form (one process: no barrier to take a maximum over)."""
    import ctypes as C
    import torch  # noqa: F401  (first: the HIP runtime torch loads is then shared with the libraries)
    import seekstorm_amd as S
    from seekstorm_amd import _native as N
    from oracle import oracle as O
    n_sh = args.one_process
    if torch.cuda.device_count() < n_sh:
        raise SystemExit(f"--one-process {n_sh}: only {torch.cuda.device_count()} GPU(s) visible")
    S.lib()
    HL = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(S.__file__)), "lib", "libseekstorm_host.so"))
    HL.ssh_index_create.restype = C.c_void_p
    HL.ssh_index_create.argtypes = [C.c_int, C.POINTER(C.c_int)]
    HL.ssh_index_destroy.argtypes = [C.c_void_p]
    HL.ssh_synth_lexical.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
    HL.ssh_index_search_lexical_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    ix = HL.ssh_index_create(n_sh, (C.c_int * n_sh)(*range(n_sh)))
    nq, k = args.queries, 10
    term_lists, th = make_c2_queries(O, nq)
    th32 = np.ascontiguousarray(th, np.uint32)
    tab = np.ascontiguousarray(O.len_table(), np.uint8)
    t0 = time.perf_counter()
    for i in range(n_sh):
        N.check(HL.ssh_synth_lexical(ix, i, int(O.LEX_SEED), int(args.docs), len(th32), th32.ctypes.data, tab.ctypes.data), "ssh_synth_lexical")
    build_s = time.perf_counter() - t0
    NB = 8
    rot = [term_lists] + [make_c2_queries(O, nq, seed=5000 + i)[0] for i in range(1, NB)]
    flats = [np.array([t for q in tl_ for t in q], np.uint32) for tl_ in rot]
    offs = [np.concatenate([[0], np.cumsum([len(q) for q in tl_])]).astype(np.uint32) for tl_ in rot]
    doc = np.empty((nq, k), np.uint64); score = np.empty((nq, k), np.float32); cnt = np.empty(nq, np.uint32); tot = np.empty(nq, np.uint64)
    it = [0]

    def call():
        b = it[0] % NB
        it[0] += 1
        N.check(HL.ssh_index_search_lexical_batch(ix, nq, flats[b].ctypes.data, offs[b].ctypes.data, int(S.QueryType.Union), k, N.RT_TOPK, 1,
                                                  doc.ctypes.data, score.ctypes.data, cnt.ctypes.data, tot.ctypes.data), "ssh_index_search_lexical_batch")

    call()
    # self-check: every query of batch 0 has k results, scores descending, global ids spread over all shards (id % S = its shard)
    assert int(cnt.min()) == k and np.all(np.diff(score, axis=1) <= 0), "merged lists are not top-k lists"
    shards_seen = np.unique(doc % n_sh)
    assert len(shards_seen) == n_sh, ("answers name docs of shards", shards_seen.tolist())
    checksum = float(score.astype(np.float64).sum())
    for _ in range(args.warmup * args.calls_per_step):
        call()
    t0 = time.perf_counter()
    for _ in range(args.steps * args.calls_per_step):
        call()
    dt = time.perf_counter() - t0
    calls = args.steps * args.calls_per_step
    line = {"metric": "queries/sec (BM25 3-term OR top-10)", "value": nq * calls / dt, "unit": "queries/s", "n_gpus": n_sh, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 sums of 16-bit posting weight codes (BM25)", "data": "synthetic",
            "config": {"workload": "C2: 10M synthetic docs per shard, 3-term OR BM25 top-10; host pointers in, merged answers on the host",
                       "docs_per_shard": args.docs, "queries_per_call": nq, "calls_per_step": args.calls_per_step, "k": k,
                       "parallelism": f"ONE process, {n_sh} shard tasks (host threads), one all-gather per call (ss_comm_create_all)",
                       "entry_point": "seekstorm_host Index::search_lexical_batch -> ss_bm25_search_sharded per shard"},
            "ms_per_call": dt / calls * 1e3, "build_s": build_s, "checksum_batch0": checksum, "shards_in_answers": shards_seen.tolist()}
    HL.ssh_index_destroy(ix)
    print_line(json.dumps(_r(line)))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--calls-per-step", type=int, default=200, help="C-ABI calls (1000-query batches) per step of the primary leg")
    ap.add_argument("--workload", default="all", choices=["all", "bm25", "vec"])
    ap.add_argument("--docs", type=int, default=10_000_000, help="docs per shard (= per GPU)")
    ap.add_argument("--rows", type=int, default=10_000_000, help="vector rows per shard")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--min-seconds", type=float, default=2.0, help="secondary legs: timed for >= this long or >= 200 batches")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-parity", action="store_true", help="skip the full-size oracle comparison")
    ap.add_argument("--no-topk-count", action="store_true", help="skip the TopkCount comparison (keeps counter profiles of the Topk kernels clean)")
    ap.add_argument("--no-rationed", action="store_true", help="skip the rationed-vocabulary leg")
    ap.add_argument("--no-fields", action="store_true", help="skip the multi-field (BM25F) leg")
    ap.add_argument("--no-vocab", action="store_true", help="skip the realistic-vocabulary leg (1 M rare terms in the sparse tier)")
    ap.add_argument("--vocab-terms", type=int, default=1_000_000)
    ap.add_argument("--scale-check", action="store_true", help="self-verification of a multi-rank run (always on when N > 1): every rank reports the ranks its "
                    "communicator spans, the microseconds of the all-gather and a checksum of the merged answers; rank 0 asserts they agree")
    ap.add_argument("--no-clustered", action="store_true", help="skip the clustered-corpus leg (10 M docs whose term densities vary with the doc's cluster)")
    ap.add_argument("--no-real-format", action="store_true", help="skip the drop-in rehearsal on a million-doc index.bin / vector.bin / delete.bin")
    ap.add_argument("--no-commit", action="store_true", help="skip the commit leg (an image with a sparse tier built level by level, tools/commit_leg.py)")
    ap.add_argument("--real-format-docs", type=int, default=1_000_000)
    ap.add_argument("--quick", action="store_true", help="profiling runs: few calls per leg, no cpu / parity legs")
    ap.add_argument("--cpu-seconds", type=float, default=5.0)
    ap.add_argument("--parity-queries", type=int, default=1000, help="C2 queries checked against the full-size oracle (all of the batch by default)")
    ap.add_argument("--cpu-queries", type=int, default=32, help="C2 queries the cpu_baseline leg cycles through")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the concurrent-callers legs (T host threads, one query per call)")
    ap.add_argument("--no-sharded", action="store_true", help="skip the ss_*_search_sharded legs at N = 1 (no RCCL communicator is then created: "
                    "profiler runs -- RCCL's initialisation faults under rocprofv3 on this image)")
    ap.add_argument("--one-process", type=int, default=0, metavar="S",
                    help="the reference's OWN shape instead of one rank per GPU: ONE process, S shards on GPUs 0 .. S-1, one host thread per shard and call "
                         "(search.rs:1637-1650), the lists exchanged over RCCL communicators made by ss_comm_create_all; prints one line of the same form")
    args = ap.parse_args()
    claim_stdout()
    if args.one_process:
        return one_process_main(args)
    if args.quick:
        args.no_cpu = args.no_parity = args.no_sharded = True
        args.calls_per_step, args.min_seconds = min(args.calls_per_step, 2), 0.0

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: start one rank per GPU ourselves (the form the driver uses for N = 1 must work for N > 1 too)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch  # first: the HIP runtime torch loads is then shared with libseekstorm_hip.so
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import seekstorm_amd as S
    from seekstorm_amd import _native as N
    from seekstorm_amd import distributed as D
    from oracle import oracle as O  # query set + thresholds (host constants); checker + cpu_baseline legs
    from oracle import fullsize as F
    L = S.lib()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # an explicit (non-null) stream: the C ABI treats a NULL stream as "the shard's own stream", and the collective must be
    # ordered after the search kernels on ONE stream
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    sptr = C.c_void_p(stream.cuda_stream)
    assert sptr.value, "expected a non-null HIP stream handle"
    sh = S.Shard(local_rank, shard_id=rank)
    sh.synth_partition(rank, world)
    # RCCL communicator behind the C ABI (ss_comm_create); at N = 1 only for the sharded legs (one rank: the exchange is the identity)
    comm = D.ShardComm(rank, world, local_rank) if (world > 1 or not args.no_sharded) else None
    merged = {}

    def timed(step_fn, steps, warmup):
        for _ in range(warmup):
            step_fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def timed_for(call_fn, min_calls=200):
        """secondary legs: >= min_calls calls and >= args.min_seconds; returns (calls, seconds)"""
        call_fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        call_fn()
        torch.cuda.synchronize()
        est = max(time.perf_counter() - t0, 1e-5)
        n = min_calls if args.min_seconds > 0 else 3
        n = int(min(max(n, args.min_seconds / est), 20000))
        if world > 1:  # every rank must issue the same number of calls: each one holds a collective
            t = torch.tensor([n], device=dev, dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            n = int(t.item())
        return n, timed(call_fn, n, 2)

    def latencies(call_fn, n):
        """device-side latency samples (HIP events on the launch stream), one sync per sample"""
        n = n if args.min_seconds > 0 else min(n, 10)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in ev:
            a.record(stream)
            call_fn()
            b.record(stream)
            torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in ev]

    def host_latencies(call_fn, n):
        """submit -> results on the host, host clock"""
        n = n if args.min_seconds > 0 else min(n, 10)
        out = []
        for _ in range(n):
            t0 = time.perf_counter()
            call_fn()
            out.append((time.perf_counter() - t0) * 1e3)
        return out

    # ------------------------------------------------------------------ BM25 (primary)
    bm = None
    parity = {}
    if args.workload in ("all", "bm25"):
        k = 10
        term_lists, th = make_c2_queries(O, args.queries)
        tab = O.len_table()
        t0 = time.perf_counter()
        sh.synth_lexical(O.LEX_SEED, args.docs, th, tab)
        build_s = time.perf_counter() - t0
        info = sh.lexical_info()
        q_np = sh.make_queries(term_lists, S.QueryType.Union)
        nq = len(q_np)
        q_dev = torch.from_numpy(q_np.view(np.uint8).reshape(nq, -1).copy()).to(dev)
        # the timed regions rotate through NB different 1000-query batches (batch 0 = the one every check refers to)
        NB = 8 if not args.quick else 2
        rot_lists = [term_lists] + [make_c2_queries(O, args.queries, seed=5000 + i)[0] for i in range(1, NB)]
        rot_np = [q_np] + [sh.make_queries(tl_, S.QueryType.Union) for tl_ in rot_lists[1:]]
        rot_dev = [q_dev] + [torch.from_numpy(qn_.view(np.uint8).reshape(nq, -1).copy()).to(dev) for qn_ in rot_np[1:]]
        rot_i = [0]
        # the host side of the primary call: the caller's own (pageable) arrays, as the seam would pass them
        hp_doc = np.empty((nq, k), np.uint32); hp_score = np.empty((nq, k), np.float32)
        hp_cnt = np.empty(nq, np.uint32); hp_tot = np.empty(nq, np.uint64)
        o_doc = torch.empty((nq, k), dtype=torch.int32, device=dev)
        o_score = torch.empty((nq, k), dtype=torch.float32, device=dev)
        o_cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
        o_tot = torch.empty((nq,), dtype=torch.int64, device=dev)
        OPS = 2 | (3 << 8)  # unions of 3 terms, no NOT terms (validated on the device by bm_expand_kernel)

        def bm_call(n=nq, rt=N.RT_TOPK, qd=None):
            N.check(L.ss_bm25_search_dev(sh._h, n, (q_dev if qd is None else qd).data_ptr(), k, rt, OPS, o_doc.data_ptr(), o_score.data_ptr(),
                                         o_cnt.data_ptr(), o_tot.data_ptr(), sptr), "ss_bm25_search_dev")
            if world > 1:  # one all-gather of the per-shard top-k over RCCL + identical merge on every rank, behind the C ABI
                merged["bm25"] = comm.allgather_merge(o_doc[:n], o_score[:n], o_cnt[:n], k, sptr)

        def bm_rot_call():
            bm_call(qd=rot_dev[rot_i[0] % NB])
            rot_i[0] += 1

        def bm_host_rot_call():
            """THE primary call: host pointers in, answers on the host (N > 1: + one all-gather and the merge, on every rank's host)"""
            qn_ = rot_np[rot_i[0] % NB]
            rot_i[0] += 1
            if world > 1:
                merged["bm25_host"] = comm.search_lexical_sharded(sh, qn_, k, N.RT_TOPK)
            else:
                N.check(L.ss_bm25_search(sh._h, nq, qn_.ctypes.data_as(C.c_void_p), k, N.RT_TOPK, N.ptr(hp_doc, N.u32p), N.ptr(hp_score, N.f32p),
                                         N.ptr(hp_cnt, N.u32p), N.ptr(hp_tot, N.u64p)), "ss_bm25_search")

        def bm_step():
            for _ in range(args.calls_per_step):
                bm_host_rot_call()

        # exact union sizes for the roofline's "1 B per scored candidate" term: one untimed TopkCount pass (exhaustive scan)
        sh.set_strategy(N.BM25_EXHAUSTIVE)
        N.check(L.ss_bm25_search_dev(sh._h, nq, q_dev.data_ptr(), k, N.RT_TOPKCOUNT, OPS, o_doc.data_ptr(),
                                     o_score.data_ptr(), o_cnt.data_ptr(), o_tot.data_ptr(), sptr), "ss_bm25_search_dev")
        torch.cuda.synchronize()
        tot = o_tot.cpu().numpy().astype(np.int64)
        ref_scores = o_score.cpu().numpy().copy()
        ref_docs = o_doc.cpu().numpy().copy()
        # algorithmic bytes: sum_t df_t*(2B id + 1B tf) + 1B per scored candidate + 4B per (term, block) + 8B*k
        uniq = sorted({t for tl_ in rot_lists for tl in tl_ for t in tl})
        dfm = dict(zip(uniq, (int(x) for x in sh.posting_count(uniq))))
        n_blocks = (args.docs + 65535) // 65536

        def batch_bytes(tls, totals):
            return np.array([sum(dfm[t] for t in tl) * 3 + int(totals[i]) + 4 * n_blocks * len(tl) + 8 * k for i, tl in enumerate(tls)], np.float64)
        bytes_q = batch_bytes(term_lists, tot)
        rot_bytes = [float(bytes_q.sum())]
        for b in range(1, NB):  # exact union sizes of the other batches of the rotation: one untimed TopkCount pass each
            bm_call(rt=N.RT_TOPKCOUNT, qd=rot_dev[b])
            torch.cuda.synchronize()
            rot_bytes.append(float(batch_bytes(rot_lists[b], o_tot.cpu().numpy().astype(np.int64)).sum()))
        bytes_launch = float(np.mean(rot_bytes))  # the timed launches rotate through the NB batches evenly

        def check_strategy(strategy):
            sh.set_strategy(strategy)
            bm_call()
            torch.cuda.synchronize()
            assert np.array_equal(ref_scores, o_score.cpu().numpy()), "Topk ranking differs from the exhaustive TopkCount pass"

        # (1) exhaustive scan: every posting of every query term is read -- the kernel the HBM roofline is about
        check_strategy(N.BM25_EXHAUSTIVE)
        sh.profile(True)
        sh.profile_read(0, reset=True)
        ex_n, ex_dt = timed_for(bm_rot_call)
        ex_launches, ex_kms = sh.profile_read(0, reset=True)
        ex_kms /= max(ex_launches, 1)
        # HIP events are recorded around every launch while profiling is on: the warm-up launches of timed_for are averaged in
        # as well; all of them are full 1000-query launches of the same kernel.
        ex_qps, ex_ms = nq * ex_n / ex_dt, ex_dt / ex_n * 1e3
        # (2) the default strategy (AUTO): top-k unions take the pruned path (MaxScore over the probe index)
        check_strategy(N.BM25_AUTO)
        rot_i[0] = 0
        bm_host_rot_call()  # batch 0 through the host-pointer call: the answers every check refers to
        if world == 1:
            assert np.array_equal(hp_score, ref_scores), "host-pointer entry point differs from the device-pointer one"
        for _ in range(args.warmup):
            bm_step()
        sh.profile_read(0, reset=True)  # the accumulators count the timed launches only
        dt = timed(bm_step, args.steps, 0)
        launches, kms = sh.profile_read(0, reset=True)
        calls = args.steps * args.calls_per_step
        assert launches == calls, (launches, calls)
        avg_ms = kms / max(launches, 1)
        qps, ms_step = nq * calls / dt, dt / args.steps * 1e3
        # the device-resident form of the same call (ss_bm25_search_dev: queries and answers stay in HBM, asynchronous on the caller's
        # stream; N > 1: + the all-gather of the per-shard lists): what a host that keeps its batches on the device gets
        dev_n, dev_dt = timed_for(bm_rot_call)
        sh.profile_read(0, reset=True)
        sh.profile(False)
        device_resident = {"value": nq * dev_n / dev_dt, "unit": "queries/s", "ms_per_call": dev_dt / dev_n * 1e3, "calls": dev_n,
                           "entry_point": "ss_bm25_search_dev (queries and answers resident in HBM)"}

        # self-verification of the exchange: what every rank's communicator spans, what one all-gather costs, and that all
        # ranks hold the SAME merged answers.  The first real multi-rank run checks itself; at N = 1 only with --scale-check.
        scale_check = None
        if comm is not None and (world > 1 or args.scale_check):
            comm.profile(True)
            comm.profile_read()
            sh_out = comm.search_lexical_sharded(sh, q_np, k, N.RT_TOPK)  # ss_bm25_search_sharded: search, ONE all-gather, merge
            n_coll, ag_us = comm.profile_read()
            comm.profile(False)
            bm_call()
            torch.cuda.synchronize()
            m_doc, m_score, m_cnt = (merged["bm25"] if world > 1 else (o_doc.to(torch.int64), o_score, o_cnt))
            digest = hashlib.sha256(m_doc.cpu().numpy().tobytes() + m_score.cpu().numpy().tobytes() + m_cnt.cpu().numpy().tobytes()).hexdigest()[:16]
            digest_h = hashlib.sha256(sh_out[0].tobytes() + sh_out[1].tobytes() + sh_out[2].tobytes()).hexdigest()[:16]
            r_, n_, d_ = comm.info()
            mine = {"rank": rank, "comm_rank": r_, "ranks_seen": n_, "device": d_, "allgather_us": ag_us, "collectives": n_coll,
                    "merged_checksum_dev": digest, "merged_checksum_sharded": digest_h}
            print("SCALE_CHECK " + json.dumps(mine), file=sys.stderr, flush=True)
            every = [mine]
            if world > 1:
                every = [None] * world
                dist.all_gather_object(every, mine)
            if rank == 0:
                assert all(e["ranks_seen"] == world for e in every), f"a communicator does not span {world} ranks: {every}"
                assert len({e["merged_checksum_dev"] for e in every}) == 1 and len({e["merged_checksum_sharded"] for e in every}) == 1, \
                    f"ranks hold different merged answers: {every}"
                scale_check = {"ranks": every, "agree": True}

        # (3) ResultType::TopkCount, the reference server's default: pruned top-k + exact union counts (popcounts over the
        # probe index's bit records) under AUTO, against the exhaustive scan in count mode
        tc = {}
        for name, strat in (() if args.no_topk_count else (("exhaustive", N.BM25_EXHAUSTIVE), ("auto", N.BM25_AUTO))):
            sh.set_strategy(strat)
            bm_call(rt=N.RT_TOPKCOUNT)
            torch.cuda.synchronize()
            assert np.array_equal(tot, o_tot.cpu().numpy().astype(np.int64)), "TopkCount totals differ between strategies"
            assert np.array_equal(ref_scores, o_score.cpu().numpy())
            n_, d_ = timed_for(lambda: bm_call(rt=N.RT_TOPKCOUNT))
            tc[name] = {"value": nq * n_ / d_, "unit": "queries/s", "ms_per_call": d_ / n_ * 1e3, "calls": n_}
        sh.set_strategy(N.BM25_AUTO)
        # (3b) the reference server's default query type: intersections.  C1-shaped 2-term AND top-10 (BASELINE.json configs[0]) on
        # this 10 M-doc corpus, one term from the 1-5 % band and one from the 5-20 % band, ResultType::TopkCount (exact counts)
        inter = None
        if not args.no_topk_count:
            rng_i = np.random.default_rng(4321)
            ba, bb = band_terms(th, 0.01, 0.05), band_terms(th, 0.05, 0.20)
            qi_np = sh.make_queries([[int(rng_i.choice(ba)), int(rng_i.choice(bb))] for _ in range(nq)], S.QueryType.Intersection)
            qi_dev = torch.from_numpy(qi_np.view(np.uint8).reshape(nq, -1).copy()).to(dev)
            OPS_AND = 1 | 64 | (2 << 8) | (2 << 16)  # intersections, every query with exactly 2 terms

            def and_call(rt=N.RT_TOPKCOUNT):
                N.check(L.ss_bm25_search_dev(sh._h, nq, qi_dev.data_ptr(), k, rt, OPS_AND, o_doc.data_ptr(), o_score.data_ptr(),
                                             o_cnt.data_ptr(), o_tot.data_ptr(), sptr), "ss_bm25_search_dev")
            sh.set_strategy(N.BM25_EXHAUSTIVE)
            and_call()
            torch.cuda.synchronize()
            and_ref = (o_score.cpu().numpy().copy(), o_tot.cpu().numpy().copy())
            sh.set_strategy(N.BM25_AUTO)
            and_call()
            torch.cuda.synchronize()
            assert np.array_equal(and_ref[0], o_score.cpu().numpy()) and np.array_equal(and_ref[1], o_tot.cpu().numpy()), "AND: strategies differ"
            n_, d_ = timed_for(and_call)
            sh.set_strategy(N.BM25_EXHAUSTIVE)
            sh.profile(True)
            sh.profile_read(0, reset=True)
            nx_, dx_ = timed_for(and_call)
            xl_, xms_ = sh.profile_read(0, reset=True)
            sh.profile(False)
            sh.set_strategy(N.BM25_AUTO)
            and_u = sorted({int(t) for q_ in qi_np["term"][:, :2] for t in q_})
            and_df = dict(zip(and_u, (int(x) for x in sh.posting_count(and_u))))
            and_bytes = float(sum((and_df[int(a_)] + and_df[int(b_)]) * 3 + int(m_) + 4 * ((args.docs + 65535) // 65536) * 2 + 8 * k
                                  for (a_, b_), m_ in zip(qi_np["term"][:, :2], and_ref[1])))
            xk_ = xms_ / max(xl_, 1)
            inter = {"value": nq * n_ / d_, "unit": "queries/s", "ms_per_call": d_ / n_ * 1e3, "calls": n_, "result_type": "TopkCount",
                     "exhaustive": {"value": nq * nx_ / dx_, "unit": "queries/s", "ms_per_call": dx_ / nx_ * 1e3, "calls": nx_,
                                    "roofline": {"bound": "hbm", "kernel": "bm25_scan16_kernel<2,1,count,AND> (entries with a level: only the first term creates them)",
                                                 "achieved": and_bytes / (xk_ * 1e-3) / 1e9 if xk_ > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                 "frac": and_bytes / (xk_ * 1e-3) / 1e9 / HBM_PEAK_GBS if xk_ > 0 else None,
                                                 "algorithmic_bytes_per_launch": and_bytes, "avg_launch_ms": xk_, "launches": int(xl_)}},
                     "workload": "2-term AND top-10, terms from the 1-5 % and 5-20 % df bands, 1000 queries per call, AUTO (pruned: the shorter "
                                 "list drives, the other is probed; counts are a by-product)", "mean_matches": float(and_ref[1].mean())}
        # (3c) the exhaustive strategy under EXCLUSIONS: C2's queries + one NOT term each (2-5 % band), 1 % of the docs tombstoned,
        # ResultType::TopkCount -- the 16-bit tile's EXCL instance (the NOT list streamed beside the terms, the sub-block's tombstone
        # words per item) against the f32 tile that used to take every such request (SS_BM25_EXHAUSTIVE_F32)
        excl = None
        if not args.no_topk_count and world == 1 and not args.quick:
            sx = S.Shard(local_rank, shard_id=rank)
            sx.synth_partition(rank, world)
            sx.synth_lexical(O.LEX_SEED, args.docs, th, tab)
            rng_x = np.random.default_rng(2468)
            gone_x = np.unique(rng_x.integers(0, args.docs, args.docs // 100, dtype=np.uint64))
            sx.set_deleted(gone_x)
            nband = band_terms(th, 0.02, 0.05)
            not_lists = []
            for tl in term_lists:
                t_ = int(rng_x.choice(nband))
                while t_ in tl:
                    t_ = int(rng_x.choice(nband))
                not_lists.append([t_])
            qx_np = sx.make_queries(term_lists, S.QueryType.Union, not_lists)
            qx_dev = torch.from_numpy(qx_np.view(np.uint8).reshape(nq, -1).copy()).to(dev)
            OPS_X = 2 | (4 << 8) | (3 << 16) | (1 << 24)  # unions of 3 terms + 1 NOT term (bits 24..27: the most NOT terms of any query)

            def x_call(rt=N.RT_TOPKCOUNT):
                N.check(L.ss_bm25_search_dev(sx._h, nq, qx_dev.data_ptr(), k, rt, OPS_X, o_doc.data_ptr(), o_score.data_ptr(),
                                             o_cnt.data_ptr(), o_tot.data_ptr(), sptr), "ss_bm25_search_dev")
            xres, xt = {}, {}
            for name, strat in (("scan16", N.BM25_EXHAUSTIVE), ("f32_tile", N.BM25_EXHAUSTIVE_F32), ("auto", N.BM25_AUTO)):
                sx.set_strategy(strat)
                x_call()
                torch.cuda.synchronize()
                xres[name] = (o_doc.cpu().numpy().copy(), o_score.cpu().numpy().copy(), o_tot.cpu().numpy().astype(np.int64))
                sx.profile(True)
                sx.profile_read(0, reset=True)
                n_, d_ = timed_for(x_call, min_calls=50 if name == "f32_tile" else 200)
                xl_, xms_ = sx.profile_read(0, reset=True)
                sx.profile(False)
                xt[name] = (nq * n_ / d_, d_ / n_ * 1e3, n_, xms_ / max(xl_, 1), int(xl_))
            for name in ("f32_tile", "auto"):
                assert np.array_equal(xres["scan16"][1], xres[name][1]) and np.array_equal(xres["scan16"][2], xres[name][2]), \
                    f"NOT + tombstones: 16-bit tile differs from {name}"
            assert not np.isin(xres["scan16"][0].astype(np.uint64), gone_x).any(), "a tombstoned doc was ranked"
            xu = sorted({t for tl in term_lists for t in tl} | {t for nl in not_lists for t in nl})
            xdf = dict(zip(xu, (int(x) for x in sx.posting_count(xu))))
            # algorithmic bytes: SURVEY 8d's formula for the scored lists + 2 B doc id per NOT posting + the tombstone bitmap (1 bit per doc)
            x_bytes = float(sum(sum(xdf[t] for t in tl) * 3 + xdf[nl[0]] * 2 + int(m_) + args.docs // 8 + 4 * n_blocks * 4 + 8 * k
                                for tl, nl, m_ in zip(term_lists, not_lists, xres["scan16"][2])))

            def x_roof(name, kernel):
                ms_ = xt[name][3]
                return {"bound": "hbm", "kernel": kernel, "achieved": x_bytes / (ms_ * 1e-3) / 1e9 if ms_ > 0 else None, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": x_bytes / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_ > 0 else None,
                        "algorithmic_bytes_per_launch": x_bytes, "avg_launch_ms": ms_, "launches": xt[name][4]}
            excl = {"value": xt["scan16"][0], "unit": "queries/s", "ms_per_call": xt["scan16"][1], "calls": xt["scan16"][2], "result_type": "TopkCount",
                    "roofline": x_roof("scan16", "bm25_scan16_kernel<3,1,count,OR,EXCL> (NOT list streamed, tombstone words per item, first-touch counts corrected)"),
                    "f32_tile": {"value": xt["f32_tile"][0], "unit": "queries/s", "ms_per_call": xt["f32_tile"][1], "calls": xt["f32_tile"][2],
                                 "roofline": x_roof("f32_tile", "bm25_scan_fast_kernel<4> (SS_BM25_EXHAUSTIVE_F32: what served this request before round 4)")},
                    "auto": {"value": xt["auto"][0], "unit": "queries/s", "ms_per_call": xt["auto"][1], "calls": xt["auto"][2]},
                    "deleted_docs": int(len(gone_x)), "mean_total": float(xres["scan16"][2].mean()),
                    "workload": "the C2 batch + one NOT term per query (df 2-5 %), 1 % of the docs tombstoned, TopkCount; scores and totals of the "
                                "three strategies asserted identical"}
            if not args.no_parity:
                t0 = time.perf_counter()
                nsx = min(128, nq)
                xans, _, _ = F.c2_answers(args.docs, term_lists[:nsx], th, k, O.OP_OR, O.RT_TOPKCOUNT, part=(rank, world),
                                          not_lists=not_lists[:nsx], deleted=gone_x)
                for i in range(nsx):
                    od, os_, otot = xans[i]
                    assert int(xres["scan16"][2][i]) == otot, f"NOT + tombstones: count of query {i}: {int(xres['scan16'][2][i])} vs oracle {otot}"
                    F.check_topk(xres["scan16"][0][i], xres["scan16"][1][i], od, os_, 1e-4, f"NOT + tombstones, query {i}")
                parity["not_tombstones"] = {"queries": nsx, "checked": "exact result_count_total, top-10 ids outside the tie band, scores rtol 1e-4; oracle = "
                                            "so_search_lex_ref with not_query_list and delete_hashset on the host-regenerated shard",
                                            "seconds": time.perf_counter() - t0}
            sx.close()
        # (3c') unions of MANY terms (11..32 terms go to union_blockid -> union_scan_32 in the reference, search.rs:3497-3520):
        # 64 queries of 16 terms each -- 12 from the 0.5-5 % df bands, 4 from the 5-15 % band -- top-10, on the C2 corpus.  AUTO runs the
        # 16-bit many-list scan (bm25_scan16m_kernel; exact counts from the probe index's bit records), against the f32 tile that served
        # such a request before (SS_BM25_EXHAUSTIVE_F32).  Roofline: SURVEY 8d's bytes from the exact df / union sizes.
        many = None
        if not args.no_topk_count and world == 1 and not args.quick:
            rng_m = np.random.default_rng(1357)
            lo_b, hi_b = band_terms(th, 0.005, 0.05), band_terms(th, 0.05, 0.15)
            nqm, ntm = 64, 16
            m_lists = [[int(x) for x in rng_m.choice(lo_b, 12, replace=False)] + [int(x) for x in rng_m.choice(hi_b, 4, replace=False)] for _ in range(nqm)]
            qm_np = sh.make_queries(m_lists, S.QueryType.Union)
            qm_dev = torch.from_numpy(qm_np.view(np.uint8).reshape(nqm, -1).copy()).to(dev)
            OPS_M = 2 | (ntm << 8) | (ntm << 16)

            def m_call(rt=N.RT_TOPK):
                N.check(L.ss_bm25_search_dev(sh._h, nqm, qm_dev.data_ptr(), k, rt, OPS_M, o_doc.data_ptr(), o_score.data_ptr(),
                                             o_cnt.data_ptr(), o_tot.data_ptr(), sptr), "ss_bm25_search_dev")
            mres, mt = {}, {}
            for name, strat in (("scan16m", N.BM25_AUTO), ("f32_tile", N.BM25_EXHAUSTIVE_F32)):
                sh.set_strategy(strat)
                m_call(N.RT_TOPKCOUNT)
                torch.cuda.synchronize()
                mres[name] = (o_doc[:nqm].cpu().numpy().copy(), o_score[:nqm].cpu().numpy().copy(), o_tot[:nqm].cpu().numpy().astype(np.int64))
                sh.profile(True)
                sh.profile_read(0, reset=True)
                n_, d_ = timed_for(m_call, min_calls=30 if name == "f32_tile" else 100)
                ml_, mms_ = sh.profile_read(0, reset=True)
                sh.profile(False)
                mt[name] = (nqm * n_ / d_, d_ / n_ * 1e3, n_, mms_ / max(ml_, 1), int(ml_))
            sh.set_strategy(N.BM25_AUTO)
            assert np.array_equal(mres["scan16m"][1], mres["f32_tile"][1]) and np.array_equal(mres["scan16m"][2], mres["f32_tile"][2]), "16-term unions: the two tiles differ"
            mu = sorted({t for tl in m_lists for t in tl})
            mdf = dict(zip(mu, (int(x) for x in sh.posting_count(mu))))
            m_bytes = float(sum(sum(mdf[t] for t in tl) * 3 + int(u_) + 4 * n_blocks * ntm + 8 * k for tl, u_ in zip(m_lists, mres["scan16m"][2])))

            def m_roof(name, kernel):
                ms_ = mt[name][3]
                return {"bound": "hbm", "kernel": kernel, "achieved": m_bytes / (ms_ * 1e-3) / 1e9 if ms_ > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": m_bytes / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_ > 0 else None, "algorithmic_bytes_per_launch": m_bytes,
                        "avg_launch_ms": ms_, "launches": mt[name][4]}
            many = {"value": mt["scan16m"][0], "unit": "queries/s", "ms_per_call": mt["scan16m"][1], "calls": mt["scan16m"][2], "terms_per_query": ntm,
                    "queries_per_call": nqm, "mean_union": float(mres["scan16m"][2].mean()), "mean_postings_per_query": float(np.mean([sum(mdf[t] for t in tl) for tl in m_lists])),
                    "roofline": m_roof("scan16m", "bm25_scan16m_kernel<1> (16 lists as a run-time loop; Topk -- the timed call; counts: bit records)"),
                    "f32_tile": {"value": mt["f32_tile"][0], "ms_per_call": mt["f32_tile"][1], "calls": mt["f32_tile"][2],
                                 "roofline": m_roof("f32_tile", "bm25_scan_group_kernel (SS_BM25_EXHAUSTIVE_F32: what served > 6 lists before round 5)")},
                    "workload": "64 unions of 16 terms (12 from the 0.5-5 % df bands, 4 from 5-15 %), top-10, C2 corpus; scores and exact counts of the two strategies asserted identical"}
            if not args.no_parity:
                t0 = time.perf_counter()
                nsm = 16
                mans = F.c2_answers_chunked(args.docs, m_lists[:nsm], th, k, O.OP_OR, O.RT_TOPKCOUNT, part=(rank, world), chunk=4)
                for i in range(nsm):
                    od, os_, otot = mans[i]
                    assert int(mres["scan16m"][2][i]) == otot, f"16-term unions: count of query {i}: {int(mres['scan16m'][2][i])} vs oracle {otot}"
                    F.check_topk(mres["scan16m"][0][i], mres["scan16m"][1][i], od, os_, 1e-4, f"16-term unions, query {i}")
                parity["union16"] = {"queries": nsm, "checked": "unions of 16 terms at full size: exact result_count_total, top-10 ids outside the tie band, scores rtol 1e-4; "
                                     "oracle = so_search_lex_ref (> 10 terms: union_scan's table walk, search_or)", "seconds": time.perf_counter() - t0}
        # (3c'') DEEP PAGES (search.rs:1658-1659: the crate's offset + length is unbounded; SS_MAX_K = 1024 per pass behind the ABI): one
        # C2 query -- host pointers in, the whole page on the host -- at k = 1024 (one pass), 4096, 16384; full-size parity at k = 4096.
        deep = None
        if not args.no_topk_count and world == 1 and not args.quick:
            qd_np = sh.make_queries(term_lists[:4], S.QueryType.Union)
            dms = {}
            for kd in (1024, 4096, 16384):
                fn = lambda kd=kd: sh.search_lexical_batch(qd_np[:1], kd, S.ResultType.TopkCount, reference_shortcuts=False)
                n_, d_ = timed_for(fn, min_calls=20)
                dms[str(kd)] = d_ / n_ * 1e3
            deep = {"ms_per_query": dms, "unit": "ms", "entry_point": "ss_bm25_search (host pointers, one 3-term union of the C2 batch, TopkCount)",
                    "note": "k > SS_MAX_K = 1024: passes of 1024 under (tombstones | the docs of the earlier passes)"}
            if not args.no_parity:
                t0 = time.perf_counter()
                kd = 4096
                dd, ds, dc, dtot = sh.search_lexical_batch(qd_np, kd, S.ResultType.TopkCount, reference_shortcuts=False)
                dans, _, _ = F.c2_answers(args.docs, term_lists[:4], th, kd, O.OP_OR, O.RT_TOPKCOUNT, part=(rank, world))
                for i in range(4):
                    od, os_, otot = dans[i]
                    assert int(dtot[i]) == otot and int(dc[i]) == len(od), f"deep page: query {i}: {int(dc[i])} of {int(dtot[i])} vs oracle {len(od)} of {otot}"
                    F.check_topk(dd[i][:int(dc[i])], ds[i][:int(dc[i])], od, os_, 1e-4, f"deep page k = {kd}, query {i}")
                parity["deep_page"] = {"queries": 4, "k": kd, "checked": "exact result_count_total, the 4096 ids outside the tie bands, scores rtol 1e-4",
                                       "seconds": time.perf_counter() - t0}
        # (3d) a CLUSTERED corpus at full size (every full-size corpus was uniform-random): the same 10 M docs / 4096
        # terms, but a term's density varies 32-fold with the doc's cluster (runs of 1024 / 8192 doc ids; oracle so_lex_cluster_thresh,
        # device lex_cluster_thresh) -- doc ids in bursts, uneven block maxima, sub-blocks a term skips entirely.  Same queries.
        clustered = None
        if not args.no_topk_count and world == 1 and not args.quick and not args.no_clustered:
            sc = S.Shard(local_rank, shard_id=rank)
            sc.synth_partition(rank, world)
            sc.synth_lexical(O.LEX_SEED_CLUSTERED, args.docs, th, tab)
            qc_np = sc.make_queries(term_lists, S.QueryType.Union)
            qc_dev = torch.from_numpy(qc_np.view(np.uint8).reshape(nq, -1).copy()).to(dev)

            def c_call(rt=N.RT_TOPK):
                N.check(L.ss_bm25_search_dev(sc._h, nq, qc_dev.data_ptr(), k, rt, OPS, o_doc.data_ptr(), o_score.data_ptr(),
                                             o_cnt.data_ptr(), o_tot.data_ptr(), sptr), "ss_bm25_search_dev")
            cres, ct = {}, {}
            for name, strat, rt_ in (("exhaustive", N.BM25_EXHAUSTIVE, N.RT_TOPKCOUNT), ("auto", N.BM25_AUTO, N.RT_TOPKCOUNT), ("auto_topk", N.BM25_AUTO, N.RT_TOPK),
                                     ("exhaustive_topk", N.BM25_EXHAUSTIVE, N.RT_TOPK)):
                sc.set_strategy(strat)
                c_call(rt_)
                torch.cuda.synchronize()
                cres[name] = (o_doc.cpu().numpy().copy(), o_score.cpu().numpy().copy(), o_tot.cpu().numpy().astype(np.int64))
                sc.profile(True)
                sc.profile_read(0, reset=True)
                n_, d_ = timed_for(lambda: c_call(rt_))
                cl_, cms_ = sc.profile_read(0, reset=True)
                sc.profile(False)
                ct[name] = {"value": nq * n_ / d_, "unit": "queries/s", "ms_per_call": d_ / n_ * 1e3, "calls": n_, "kernel_ms": cms_ / max(cl_, 1)}
            assert np.array_equal(cres["exhaustive"][1], cres["auto"][1]) and np.array_equal(cres["exhaustive"][2], cres["auto"][2]), "clustered corpus: strategies differ"
            assert np.array_equal(cres["exhaustive"][1], cres["auto_topk"][1]) and np.array_equal(cres["exhaustive"][1], cres["exhaustive_topk"][1])
            cu = sorted({t for tl in term_lists for t in tl})
            cdf = dict(zip(cu, (int(x) for x in sc.posting_count(cu))))
            c_bytes = float(sum(sum(cdf[t] for t in tl) * 3 + int(m_) + 4 * n_blocks * len(tl) + 8 * k for tl, m_ in zip(term_lists, cres["exhaustive"][2])))
            clustered = dict(ct, docs=args.docs, mean_union=float(cres["exhaustive"][2].mean()), mean_df_ratio_to_uniform=float(np.mean([cdf[t] / max(dfm.get(t, 1), 1) for t in cu if t in dfm])),
                             exhaustive_roofline={"bound": "hbm", "kernel": "bm25_scan16_kernel<3,1> on the clustered corpus", "algorithmic_bytes_per_launch": c_bytes,
                                                  "avg_launch_ms": ct["exhaustive_topk"]["kernel_ms"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                  "achieved": c_bytes / (ct["exhaustive_topk"]["kernel_ms"] * 1e-3) / 1e9 if ct["exhaustive_topk"]["kernel_ms"] > 0 else None,
                                                  "frac": c_bytes / (ct["exhaustive_topk"]["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if ct["exhaustive_topk"]["kernel_ms"] > 0 else None},
                             pruned_effective_GBs_on_algorithmic_bytes=c_bytes / (ct["auto_topk"]["kernel_ms"] * 1e-3) / 1e9 if ct["auto_topk"]["kernel_ms"] > 0 else None,
                             workload="the C2 batch (3-term OR top-10, 1000 queries per call) on a 10 M-doc corpus whose term densities vary 32-fold with the doc's "
                                      "cluster (seed bit 63); exhaustive / auto = TopkCount, *_topk = Topk; answers of all four asserted identical")
            if not args.no_parity:
                t0 = time.perf_counter()
                nsc = min(128, nq)
                cans = F.c2_answers_chunked(args.docs, term_lists[:nsc], th, k, O.OP_OR, O.RT_TOPKCOUNT, seed=O.LEX_SEED_CLUSTERED, part=(rank, world), chunk=64)
                for i in range(nsc):
                    od, os_, otot = cans[i]
                    assert int(cres["auto"][2][i]) == otot, f"clustered corpus: count of query {i}: {int(cres['auto'][2][i])} vs oracle {otot}"
                    F.check_topk(cres["auto"][0][i], cres["auto"][1][i], od, os_, 1e-4, f"clustered corpus, query {i}")
                parity["clustered"] = {"queries": nsc, "docs": args.docs, "checked": "exact result_count_total, top-10 ids outside the tie band, scores rtol 1e-4; "
                                       "oracle = so_search_lex_ref on the host-regenerated clustered shard (Rle / Bitmap / Array containers by the reference's chooser)",
                                       "seconds": time.perf_counter() - t0}
            sc.close()
        ach_alg = bytes_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        ex_ach = bytes_launch / (ex_kms * 1e-3) / 1e9 if ex_kms > 0 else 0.0
        lat_batch = latencies(bm_rot_call, 1000)
        lat_one = latencies(lambda: bm_call(1), 1000)

        # (4) end to end through the host-pointer entry point: H2D of the queries + kernels + D2H of the results + sync
        h_doc = np.empty((nq, k), np.uint32); h_score = np.empty((nq, k), np.float32)
        h_cnt = np.empty(nq, np.uint32); h_tot = np.empty(nq, np.uint64)

        def bm_host_call(n=nq):
            N.check(L.ss_bm25_search(sh._h, n, q_np.ctypes.data_as(C.c_void_p), k, N.RT_TOPK, N.ptr(h_doc, N.u32p), N.ptr(h_score, N.f32p),
                                     N.ptr(h_cnt, N.u32p), N.ptr(h_tot, N.u64p)), "ss_bm25_search")
        bm_host_call()
        assert np.array_equal(h_score, ref_scores), "host-pointer entry point differs from the device-pointer one"
        e2e_lat = host_latencies(bm_host_call, 1000)
        e2e_one = host_latencies(lambda: bm_host_call(1), 1000)
        # small host-pointer batches (the reference's call shape is ONE query per call): <= 64 queries take the one-launch path of
        # bm25_small.hip -- queries in the kernel arguments, answers + completion flag written straight into pinned host memory
        small = {}
        for n_ in (8, 32, 64):
            if n_ <= nq:
                assert np.array_equal(h_score[:n_], ref_scores[:n_])
                l_ = host_latencies(lambda: bm_host_call(n_), 400)
                assert np.array_equal(h_score[:n_], ref_scores[:n_]), "one-launch path differs from the staged pipeline"
                small[f"nq{n_}_ms_p50"] = pct(l_, 50); small[f"nq{n_}_ms_p99"] = pct(l_, 99); small[f"nq{n_}_qps"] = n_ / (np.mean(l_) * 1e-3)
        end_to_end = {"value": nq / (np.mean(e2e_lat) * 1e-3), "unit": "queries/s", "entry_point": "ss_bm25_search (host pointers: H2D queries, "
                      "kernels, D2H results, sync; host clock)", "batch_ms_p50": pct(e2e_lat, 50), "batch_ms_p99": pct(e2e_lat, 99),
                      "single_query_ms_p50": pct(e2e_one, 50), "single_query_ms_p99": pct(e2e_one, 99), "samples": len(e2e_lat),
                      "small_batches": small, "small_batches_path": "bm25_small_kernel: ONE launch per call of <= 64 queries (SS_BM25_SMALL=0: the staged pipeline)"}

        # Roofline of the dominant kernel of the timed region (bm25_probe_kernel).  A pruning kernel answers WITHOUT reading
        # most of SURVEY 8d's algorithmic bytes, so algorithmic bytes / time exceeds the HBM peak and says nothing about it:
        # `achieved` is the traffic the counters saw (profiles/pmc_traffic.json, refused when collected on other kernel
        # sources; counter -> byte conversion calibrated on this access pattern, tools/probes/pmc_calib.hip) over the live
        # launch time.  The exhaustive scan below is the kernel SURVEY 8d's figure applies to.
        moved = pmc_traffic("bm25_pruned")
        moved_lo = pmc_traffic("bm25_pruned", "hbm_bytes_per_launch_low")
        real = moved / (avg_ms * 1e-3) / 1e9 if (moved and avg_ms > 0) else None
        bm = dict(qps=qps, ms_per_step=ms_step, build_s=build_s, info=info, device_resident=device_resident,
                  roofline={"bound": "hbm", "kernel": "bm25_probe_kernel<3,1> (pruned strategy: essential terms' postings + probe records)",
                            "achieved": real, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (real / HBM_PEAK_GBS) if real else None,
                            "frac_counter": (real / HBM_PEAK_GBS) if real else None,
                            "frac_counter_low": (moved_lo / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (moved_lo and avg_ms > 0) else None,
                            "traffic": moved, "algorithmic_bytes_per_launch": bytes_launch,
                            "effective_GBs_on_algorithmic_bytes": ach_alg, "avg_launch_ms": avg_ms, "launches": int(launches),
                            # SURVEY 8(d)'s own figure -- algorithmic bytes / live launch time -- for the kernels it applies to (the exhaustive
                            # strategy streams every posting), inside `roofline` so that the driver's record keeps them:
                            "sec8d_kernel": "bm25_scan16_kernel<3,1> (exhaustive strategy on the same batches)",
                            "sec8d_frac": ex_ach / HBM_PEAK_GBS, "sec8d_achieved_GBs": ex_ach, "sec8d_avg_launch_ms": ex_kms,
                            "sec8d_algorithmic_bytes": bytes_launch, "sec8d_launches": int(ex_launches), "sec8d_traffic": pmc_traffic("bm25"),
                            "and_exhaustive_frac": ((inter or {}).get("exhaustive", {}).get("roofline", {}) or {}).get("frac"),
                            "and_exhaustive_avg_launch_ms": ((inter or {}).get("exhaustive", {}).get("roofline", {}) or {}).get("avg_launch_ms"),
                            "and_exhaustive_algorithmic_bytes": ((inter or {}).get("exhaustive", {}).get("roofline", {}) or {}).get("algorithmic_bytes_per_launch"),
                            "not_tombstones_frac": ((excl or {}).get("roofline") or {}).get("frac"),
                            "not_tombstones_avg_launch_ms": ((excl or {}).get("roofline") or {}).get("avg_launch_ms"),
                            "not_tombstones_algorithmic_bytes": ((excl or {}).get("roofline") or {}).get("algorithmic_bytes_per_launch"),
                            "fallback_f32_frac": (((excl or {}).get("f32_tile") or {}).get("roofline") or {}).get("frac"),
                            "fallback_f32_avg_launch_ms": (((excl or {}).get("f32_tile") or {}).get("roofline") or {}).get("avg_launch_ms"),
                            "clustered_exhaustive_frac": ((clustered or {}).get("exhaustive_roofline") or {}).get("frac"),
                            "clustered_exhaustive_avg_launch_ms": ((clustered or {}).get("exhaustive_roofline") or {}).get("avg_launch_ms"),
                            "pmc_profile": "profiles/pmc_traffic.json" if moved else "absent or collected on other kernel sources (stale): no counter figure",
                            "note": "achieved = counter-measured HBM bytes of this kernel / live kernel time (the kernel prunes: it answers "
                                    "without reading most of SURVEY 8d's algorithmic bytes, so effective_GBs_on_algorithmic_bytes may exceed "
                                    "the peak); exhaustive.roofline is the SURVEY 8d figure"},
                  exhaustive={"value": ex_qps, "unit": "queries/s", "ms_per_call": ex_ms, "calls": ex_n,
                              "roofline": {"bound": "hbm", "kernel": "bm25_scan16_kernel<3,1> (16-bit bound accumulators + exact re-scoring of the candidates)", "achieved": ex_ach,
                                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ex_ach / HBM_PEAK_GBS,
                                           "traffic": pmc_traffic("bm25"), "algorithmic_bytes_per_launch": bytes_launch,
                                           "avg_launch_ms": ex_kms, "launches": int(ex_launches)}},
                  topk_count=dict(tc, note="same batch with ResultType::TopkCount (exact result_count_total, the reference server's "
                                           "default): AUTO = pruned top-k + union counts from the probe index's bit records"),
                  latency_ms={"batch_p50": pct(lat_batch, 50), "batch_p99": pct(lat_batch, 99), "batch_samples": len(lat_batch),
                              "single_query_p50": pct(lat_one, 50), "single_query_p99": pct(lat_one, 99),
                              "single_query_samples": len(lat_one), "clock": "HIP events on the launch stream (device resident)"},
                  end_to_end=end_to_end, intersection=inter, exhaustive_not_tombstones=excl, clustered=clustered, scale_check=scale_check, union16=many, deep_page=deep,
                  mean_bytes_per_query=float(bytes_q.mean()), mean_union=float(tot.mean()))
        # correctness guard inside the bench: sorted, k results
        bm_call()
        torch.cuda.synchronize()
        sc = o_score.cpu().numpy()
        assert np.all(sc[:, :-1] >= sc[:, 1:]) and np.all(o_cnt.cpu().numpy() == k)

        # (4b) SEVERAL indexed fields (BM25F; the reference's README benchmarks index title / body / url): a host-built 2 M-doc corpus
        # with 3 fields (boosts 2 / 1 / 0.5), 96 terms in the C2 df bands; 2-term AND and 3-term OR top-10 with exact counts, 1000
        # queries per call through the host-pointer entry point.  Queries without a field filter read the merged per-term lists.
        if rank == 0 and not args.quick and not args.no_fields:
            t0 = time.perf_counter()
            rng_f = np.random.default_rng(7)
            fd, ff_n, ft_n = 2_000_000, 3, 96
            flens = np.clip(np.round(np.exp(np.log([12, 300, 8])[:, None] + 0.5 * rng_f.standard_normal((3, fd)))), 1, 60000).astype(np.int64)
            lut = np.array([O.lib().so_int_to_byte4(int(x)) for x in range(0, 60001)], np.uint8)
            fdl = lut[flens]
            pf = np.array([0.25, 0.9, 0.15])
            fdfs = np.concatenate([rng_f.uniform(0.005, 0.02, 32), rng_f.uniform(0.02, 0.05, 32), rng_f.uniform(0.05, 0.15, 32)])
            foffs, FD, FF, FT = [0], [], [], []
            for df_ in fdfs:
                d_ = np.sort(rng_f.choice(fd, int(df_ * fd), replace=False)).astype(np.uint32)
                m_ = rng_f.random((len(d_), ff_n)) < pf
                m_[~m_.any(1), 1] = True
                di_, fi_ = np.nonzero(m_)
                FD.append(d_[di_]); FF.append(fi_.astype(np.uint8)); FT.append(np.minimum(rng_f.geometric(0.5, len(di_)), 500).astype(np.uint16))
                foffs.append(foffs[-1] + len(di_))
            sf = S.Shard(local_rank)
            sf.upload_lexical_fields(fd, fdl, [2.0, 1.0, 0.5], np.array(foffs, np.uint64), np.concatenate(FD), np.concatenate(FF), np.concatenate(FT))
            fbuild = time.perf_counter() - t0
            lo_, mid_, hi_ = np.arange(0, 32), np.arange(32, 64), np.arange(64, 96)
            legs = {}
            for name, tl_, qt_ in (("and2", [[int(rng_f.choice(mid_)), int(rng_f.choice(hi_))] for _ in range(nq)], S.QueryType.Intersection),
                                   ("or3", [[int(rng_f.choice(lo_)), int(rng_f.choice(mid_)), int(rng_f.choice(hi_))] for _ in range(nq)], S.QueryType.Union)):
                qf = sf.make_queries(tl_, qt_)
                sf.set_strategy(N.BM25_EXHAUSTIVE)
                ref_f = sf.search_lexical_batch(qf, k, S.ResultType.TopkCount)
                sf.set_strategy(N.BM25_AUTO)
                got_f = sf.search_lexical_batch(qf, k, S.ResultType.TopkCount)
                assert all(np.array_equal(x, y) for x, y in zip(ref_f, got_f)), "multi-field: strategies differ"
                lat_f = host_latencies(lambda: sf.search_lexical_batch(qf, k, S.ResultType.TopkCount), 200)
                ab_f = sf.algorithmic_bytes(qf, got_f[3], k, fd)
                legs[name] = {"value": nq / (np.mean(lat_f) * 1e-3), "unit": "queries/s", "batch_ms_p50": pct(lat_f, 50), "batch_ms_p99": pct(lat_f, 99),
                              "roofline": {"bound": "hbm", "achieved": ab_f / (np.mean(lat_f) * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": ab_f / (np.mean(lat_f) * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_call": ab_f,
                                           "clock": "host clock around the whole call (H2D, kernels, D2H); df = docs holding the term in any field"}}
            bm["multi_field"] = dict(legs, docs=fd, fields=ff_n, boosts=[2.0, 1.0, 0.5], postings=int(foffs[-1]), build_s=fbuild, result_type="TopkCount",
                                     entry_point="ss_bm25_search (host pointers, host clock; Python mirror call)",
                                     note="and2 = 2-term AND, or3 = 3-term OR, top-10, 1000 queries per call, no field filter: the queries read one "
                                          "merged list per term (weights = sum over the fields of boost * w) and take the pruned strategy; AUTO "
                                          "answers asserted equal to the exhaustive strategy's")
            sf.close()

        # (5) a RATIONED vocabulary: the probe index has rows for the longest lists only (ss_bm25_set_probe_budget), as for a
        # real vocabulary of millions of terms.  Here: a budget of one row per list with df >= 1 % of the docs, which leaves more
        # than half of the batch's queries (their rarest term comes from the 0.5-2 % band) with a list without a fixed row.  The
        # host-pointer entry point builds pool rows for the row-less lists a batch touches and runs what is still left without
        # rows on the scan kernels (mixed batch as two) -- instead of scanning everything.  Same corpus, same queries, same answers.
        if rank == 0 and not args.quick and not args.no_rationed:
            sr = S.Shard(local_rank)
            sr.synth_partition(rank, world)  # the same shard of the corpus stream as `sh`
            frac = th.astype(np.float64) / 2.0 ** 32
            n_rows = int((frac >= 0.01).sum())
            n_sub = (args.docs + 4095) // 4096
            sr.set_probe_budget((n_rows + 1) * n_sub * 64 * 12)
            sr.synth_lexical(O.LEX_SEED, args.docs, th, tab)
            fixed_q = np.array([bool(sr.terms_probed(tl).all()) for tl in term_lists])

            def r_call():
                N.check(L.ss_bm25_search(sr._h, nq, q_np.ctypes.data_as(C.c_void_p), k, N.RT_TOPK, N.ptr(h_doc, N.u32p), N.ptr(h_score, N.f32p),
                                         N.ptr(h_cnt, N.u32p), N.ptr(h_tot, N.u64p)), "ss_bm25_search")
            t0 = time.perf_counter()
            r_call()  # cold: builds pool rows for the row-less lists of the batch
            first_ms = (time.perf_counter() - t0) * 1e3
            assert np.array_equal(h_score, ref_scores), "rationed vocabulary: answers differ"
            probed_q = np.array([bool(sr.terms_probed(tl).all()) for tl in term_lists])
            r_lat = host_latencies(r_call, 200)
            # churn: eight DIFFERENT batches in turn -- every call finds most of its row-less lists evicted and rebuilds them
            churn_q = [sr.make_queries(make_c2_queries(O, args.queries, seed=777 + i)[0], S.QueryType.Union) for i in range(8)]
            churn_i = [0]

            def c_call():
                qq = churn_q[churn_i[0] % len(churn_q)]
                churn_i[0] += 1
                N.check(L.ss_bm25_search(sr._h, nq, qq.ctypes.data_as(C.c_void_p), k, N.RT_TOPK, N.ptr(h_doc, N.u32p), N.ptr(h_score, N.f32p),
                                         N.ptr(h_cnt, N.u32p), N.ptr(h_tot, N.u64p)), "ss_bm25_search")
            c_lat = host_latencies(c_call, 96)
            sr.set_strategy(N.BM25_EXHAUSTIVE)
            r_call()
            x_lat = host_latencies(r_call, 100)
            bm["rationed_vocabulary"] = {
                "value": nq / (np.mean(r_lat) * 1e-3), "unit": "queries/s", "entry_point": "ss_bm25_search (host pointers, host clock)",
                "probe_rows": n_rows, "vocabulary": int(len(th)),
                "rows_rule": "budget = one row per list with df >= 1 % of the docs; three quarters go to the longest lists, one quarter is the pool of rows built on demand",
                "queries_with_fixed_rows": float(fixed_q.mean()), "queries_with_all_rows": float(probed_q.mean()),
                "first_call_ms": first_ms, "batch_ms_p50": pct(r_lat, 50), "batch_ms_p99": pct(r_lat, 99),
                "churn_value": nq / (np.mean(c_lat) * 1e-3), "churn_batch_ms_p50": pct(c_lat, 50),
                "all_scan_value": nq / (np.mean(x_lat) * 1e-3),
                "note": "churn_value: eight different batches in turn, so that every call rebuilds most of its pool rows; value: steady state of a repeated batch: the row-less lists it touches hold pool rows after the first call (first_call_ms "
                        "includes building them); queries still left without rows run on the scan kernels (mixed batch split in the library); "
                        "all_scan_value = the same batch with every query on the scan kernels"}
            sr.close()

        # (6) a REALISTIC vocabulary: the 4096 lists above are only the head of a real index's dictionary.  --vocab-terms rare terms
        # (Zipfian tail: df = df_min_head * 4096 / rank, >= 2) go into the image's SPARSE tier (plain sorted lists, no directory / probe
        # rows; ss_bm25_append_sparse); 3-term OR queries whose terms are drawn by rank popularity (P ~ 1 / rank over the WHOLE
        # vocabulary, dense head ranked by df) -- most queries then name at least one rare term -- through the host-pointer entry
        # point, against the same queries with their rare terms dropped (what the dense image alone would answer).
        if rank == 0 and world == 1 and not args.quick and not args.no_vocab:
            t0 = time.perf_counter()
            rng_v = np.random.default_rng(99)
            n_head, n_tail = len(th), int(args.vocab_terms)
            head_df = np.array([int(x) for x in sh.posting_count(list(range(n_head)))], np.int64)
            head_by_rank = np.argsort(-head_df, kind="stable")
            df_min = max(int(head_df.min()), 2)
            ranks_tail = np.arange(n_head + 1, n_head + n_tail + 1, dtype=np.float64)
            tail_df = np.maximum(2, (df_min * n_head / ranks_tail).astype(np.int64))
            tot_tail = int(tail_df.sum())
            lid = np.repeat(np.arange(n_tail, dtype=np.uint64), tail_df)
            key = np.unique((lid << np.uint64(32)) | rng_v.integers(0, args.docs, tot_tail, dtype=np.uint64))  # sorted by (list, doc), duplicates gone
            v_docs = (key & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            v_offs = np.zeros(n_tail + 1, np.uint64)
            v_offs[1:] = np.cumsum(np.bincount((key >> np.uint64(32)).astype(np.int64), minlength=n_tail))
            v_tfs = np.minimum(rng_v.geometric(0.6, len(v_docs)), 60).astype(np.uint16)
            gen_s = time.perf_counter() - t0
            t0 = time.perf_counter()
            first_sparse = sh.append_sparse(v_offs, v_docs, v_tfs)
            append_s = time.perf_counter() - t0
            assert first_sparse == n_head
            # queries: P(rank r) ~ 1 / r over ranks 1 .. n_head + n_tail
            pr = 1.0 / np.arange(1, n_head + n_tail + 1, dtype=np.float64)
            pr /= pr.sum()
            draws = rng_v.choice(n_head + n_tail, size=(nq, 3), p=pr)
            v_lists = []
            for row in draws:
                ids = []
                for r_ in row:
                    tid = int(head_by_rank[r_]) if r_ < n_head else int(n_head + (r_ - n_head))
                    if tid not in ids:
                        ids.append(tid)
                v_lists.append(ids)
            with_sparse = np.array([any(t >= n_head for t in tl) for tl in v_lists])
            dense_only = [[t for t in tl if t < n_head] or [tl[0] % n_head] for tl in v_lists]
            qv_full = sh.make_queries(v_lists, S.QueryType.Union)
            qv_dense = sh.make_queries(dense_only, S.QueryType.Union)
            vh_doc = np.empty((nq, k), np.uint32); vh_score = np.empty((nq, k), np.float32); vh_cnt = np.empty(nq, np.uint32); vh_tot = np.empty(nq, np.uint64)

            def v_call(qq, rt=N.RT_TOPK):
                N.check(L.ss_bm25_search(sh._h, nq, qq.ctypes.data_as(C.c_void_p), k, rt, N.ptr(vh_doc, N.u32p), N.ptr(vh_score, N.f32p),
                                         N.ptr(vh_cnt, N.u32p), N.ptr(vh_tot, N.u64p)), "ss_bm25_search")
            sh.set_strategy(N.BM25_AUTO)
            v_call(qv_full, N.RT_TOPKCOUNT)
            got_v = (vh_doc.copy(), vh_score.copy(), vh_cnt.copy(), vh_tot.copy())
            lat_full = host_latencies(lambda: v_call(qv_full), 200)
            lat_dense = host_latencies(lambda: v_call(qv_dense), 200)
            n_sp, p_sp, b_sp = sh.sparse_info()
            dense_bytes = int(info["n_postings"]) * 4 + n_head * ((args.docs + 4095) // 4096 + 1) * 4
            # parity: 16 of the queries that name a rare term against the oracle holding dense and rare lists alike
            if not args.no_parity:
                pick = [i for i in range(nq) if with_sparse[i]][:16]
                voc_d = sorted({t for i in pick for t in v_lists[i] if t < n_head})
                voc_s = sorted({t for i in pick for t in v_lists[i] if t >= n_head})
                dl_h, o_h, d_h, t_h = F.c2_corpus(args.docs, voc_d, th) if voc_d else (O.lex_doclen(args.docs), np.zeros(1, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.uint16))
                so = [0]; sd = []; stf = []
                for t in voc_s:
                    a_, b_ = int(v_offs[t - n_head]), int(v_offs[t - n_head + 1])
                    sd.append(v_docs[a_:b_]); stf.append(v_tfs[a_:b_]); so.append(so[-1] + (b_ - a_))
                osh_v = O.Shard(args.docs, dl_h, np.concatenate([o_h, o_h[-1] + np.asarray(so[1:], np.uint64)]),
                                np.concatenate([d_h] + sd), np.concatenate([t_h] + stf))
                remap_v = {t: j for j, t in enumerate(voc_d + voc_s)}
                for i in pick:
                    od, os_, otot = osh_v.search_exhaustive([remap_v[t] for t in v_lists[i]], O.OP_OR, k)
                    assert int(got_v[3][i]) == otot, f"vocabulary leg: count of query {i}: {int(got_v[3][i])} vs oracle {otot}"
                    F.check_topk(got_v[0][i, :got_v[2][i]], got_v[1][i, :got_v[2][i]], od, os_, 1e-4, f"vocabulary leg, query {i}")
                parity["vocabulary"] = {"queries": len(pick), "checked": "3-term unions naming rare (sparse-tier) terms: exact counts, top-10 ids outside the tie band, scores rtol 1e-4"}
            ab_v = sh.algorithmic_bytes(qv_full, got_v[3], k, args.docs)
            bm["realistic_vocabulary"] = {
                "roofline": {"bound": "hbm", "achieved": ab_v / (np.mean(lat_full) * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ab_v / (np.mean(lat_full) * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_call": ab_v,
                             "clock": "host clock around the whole call: queries in, the dense tier's pruned kernels, the sparse kernel, the per-query merge, answers "
                                      "out.  The dense part prunes (it answers without most of these bytes), so the fraction is effective bytes per second"},
                "value": nq / (np.mean(lat_full) * 1e-3), "unit": "queries/s", "entry_point": "ss_bm25_search (host pointers, host clock), 1000 queries per call",
                "batch_ms_p50": pct(lat_full, 50), "batch_ms_p99": pct(lat_full, 99),
                "same_queries_without_their_rare_terms": nq / (np.mean(lat_dense) * 1e-3),
                "vocabulary": n_head + n_sp, "dense_terms": n_head, "sparse_terms": n_sp, "sparse_postings": p_sp,
                "sparse_tier_bytes": b_sp, "dense_postings_and_directory_bytes": dense_bytes,
                "directory_bytes_if_dense": int(n_sp) * ((args.docs + 4095) // 4096 + 1) * 4,
                "queries_naming_a_rare_term": float(with_sparse.mean()), "mean_rare_df": float(tail_df.mean()),
                "host_generation_s": gen_s, "append_s": append_s,
                "note": "terms drawn with P ~ 1 / rank over the whole vocabulary (head = the 4096 dense lists ranked by df, tail = the rare terms); "
                        "a query's dense terms run through the ordinary kernels, every doc of its rare lists is scored in full by binary-search "
                        "probes of the other lists (bm25_sparse_kernel), the two lists are merged per query; directory_bytes_if_dense = what "
                        "the rare lists' sub-block directory rows alone would cost in the dense image"}

        # (7) DROP-IN REHEARSAL: a shard's files as the reference writes them (index.bin of a text-shaped corpus with >= 1 M keys, clustered
        # doc ids, NgramFF | NgramFFF keys and positions; vector.bin; delete.bin -- oracle/ss_textindex.c is the mini indexer) opened through
        # ss_index_bin_open -> tier -> upload with positions, queried with 2-term ANDs, 3-term ORs, phrases over n-gram keys, hybrid; every
        # answer against the oracle; 64 concurrent callers through Index::search of the C++ mirror (tools/real_format.py)
        if rank == 0 and world == 1 and not args.quick and not args.no_real_format:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import real_format
            bm["real_format"] = real_format.run(n_docs=args.real_format_docs, vocab=args.real_format_docs, parity=not args.no_parity,
                                                seconds=max(args.min_seconds, 1.0))
            if "parity" in bm["real_format"]:
                parity["real_format"] = bm["real_format"].pop("parity")

        # (8) COMMITS: an image with a sparse tier built by 16 commits of one level each (dense terms + the level's rare postings), ms per
        # commit, answers == a one-shot upload of the same docs (tools/commit_leg.py).  A leg that fails reports its error and nothing else.
        if rank == 0 and world == 1 and not args.quick and not args.no_commit:
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import commit_leg
                bm["commit"] = commit_leg.run(S, O, th)
            except Exception as e:  # noqa: BLE001 -- the headline must not depend on a secondary leg
                bm["commit"] = {"error": repr(e)[:300]}

        # ---- full-size parity (C2): a sample of the batch against the oracle on the same 10 M-doc shard, regenerated on
        # the host -- doc ids outside the tie band, scores 1e-4 relative, exact result_count_total; AUTO and EXHAUSTIVE
        if rank == 0 and not args.no_parity:
            t0 = time.perf_counter()
            ns = min(args.parity_queries, nq)
            ans = F.c2_answers_chunked(args.docs, term_lists[:ns], th, k, O.OP_OR, O.RT_TOPKCOUNT, part=(rank, world))
            runs = {}
            for sname, strat in (("exhaustive", N.BM25_EXHAUSTIVE), ("auto", N.BM25_AUTO)):
                sh.set_strategy(strat)
                for rname, rt_ in (("Topk", N.RT_TOPK), ("TopkCount", N.RT_TOPKCOUNT)):
                    N.check(L.ss_bm25_search_dev(sh._h, nq, q_dev.data_ptr(), k, rt_, OPS, o_doc.data_ptr(), o_score.data_ptr(), o_cnt.data_ptr(),
                                                 o_tot.data_ptr(), sptr), "ss_bm25_search_dev")  # this shard's own lists (no exchange)
                    torch.cuda.synchronize()
                    runs[(sname, rname)] = (o_doc.cpu().numpy().copy(), o_score.cpu().numpy().copy(), o_tot.cpu().numpy().astype(np.int64))
            for (sname, rname), (gd, gs, gt) in runs.items():
                for i in range(ns):
                    od, os_, otot = ans[i]
                    F.check_topk(gd[i], gs[i], od, os_, 1e-4, f"C2 full size, {sname} {rname}, query {i}")
                    if rname == "TopkCount":
                        assert int(gt[i]) == otot, f"C2 full size: result_count_total of query {i}: {int(gt[i])} vs oracle {otot}"
            parity["c2"] = {"queries": ns, "docs": args.docs, "checked": "top-10 doc ids outside the tie band, scores rtol 1e-4, exact "
                            "result_count_total; strategies AUTO and EXHAUSTIVE x Topk and TopkCount; oracle = so_search_lex_ref "
                            "(union_docid_3) on the host-regenerated shard", "seconds": time.perf_counter() - t0}
            # 2-term intersections of the AND leg: exact counts, ids outside ties, both strategies
            if inter is not None:
                t0 = time.perf_counter()
                na = min(256, nq)
                and_lists = [[int(x) for x in qi_np["term"][i][:2]] for i in range(na)]
                aans = F.c2_answers_chunked(args.docs, and_lists, th, k, O.OP_AND, O.RT_TOPKCOUNT, part=(rank, world), chunk=128)
                for sname, strat in (("exhaustive", N.BM25_EXHAUSTIVE), ("auto", N.BM25_AUTO)):
                    sh.set_strategy(strat)
                    and_call()
                    torch.cuda.synchronize()
                    gd, gs, gc, gt = o_doc.cpu().numpy(), o_score.cpu().numpy(), o_cnt.cpu().numpy(), o_tot.cpu().numpy().astype(np.int64)
                    for i in range(na):
                        od, os_, otot = aans[i]
                        assert int(gt[i]) == otot, f"AND full size, {sname}: count of query {i}: {int(gt[i])} vs oracle {otot}"
                        F.check_topk(gd[i, :gc[i]], gs[i, :gc[i]], od, os_, 1e-4, f"AND full size, {sname}, query {i}")
                parity["and2"] = {"queries": na, "checked": "exact match counts, top-10 ids outside the tie band, scores rtol 1e-4; AUTO and EXHAUSTIVE",
                                  "seconds": time.perf_counter() - t0}
            sh.set_strategy(N.BM25_AUTO)

        if rank == 0 and world == 1 and not args.no_cpu:
            # cpu_baseline: the reference's own algorithm for this query shape -- union_docid_3's sub-query decomposition over
            # intersection_blockid / single_blockid with block-max pruning (oracle so_search_lex_ref) -- in the reference's
            # execution structure: S document-partitioned shards, one task per shard and query.  Bounded sample of the same
            # queries on the same corpus.
            from concurrent.futures import ThreadPoolExecutor
            eff_cpus, cpu_info = F.effective_cpus()
            cores = max(1, min(F.host_threads(128), int(np.ceil(eff_cpus))))  # threads spawned: one per CPU the process can really run on
            ns = min(args.cpu_queries, nq)
            _, osh, remap = F.c2_answers(args.docs, term_lists[:ns], th, k, O.OP_OR, O.RT_TOPK)
            qs = np.array([[remap[t] for t in tl] for tl in term_lists[:ns]], np.uint32)
            t0 = time.perf_counter()
            one_tp, _, _ = O.bench_lex([osh], qs, O.OP_OR, k, O.RT_TOPK, 0, cores, args.cpu_seconds)
            one_lat_qps, _, one_lat = O.bench_lex([osh], qs, O.OP_OR, k, O.RT_TOPK, 1, 1, min(args.cpu_seconds, 3.0))
            parts = O.split_corpus(args.docs, osh.doclen[:args.docs], osh.offs, osh.docs, osh.tfs, cores)
            with ThreadPoolExecutor(cores) as ex:
                shards = list(ex.map(lambda x: O.Shard(*x), parts))
            s_tp, _, _ = O.bench_lex(shards, qs, O.OP_OR, k, O.RT_TOPK, 0, cores, args.cpu_seconds)
            s_lat_qps, _, s_lat = O.bench_lex(shards, qs, O.OP_OR, k, O.RT_TOPK, 1, cores, args.cpu_seconds)
            best = max(one_tp, s_tp)
            bm["cpu_baseline"] = {
                "value": best, "unit": "queries/s", "cores": eff_cpus, "threads": cores, "cpu_info": cpu_info, "kind": "port", "algorithm": "union_docid_3",
                "cores_note": "cores = the CPUs this process can run on at once (min of logical CPUs, affinity mask, cgroup CPU quota); "
                              "threads = worker threads spawned",
                "shards": 1 if best == one_tp else cores,
                "sample": f"{ns} of the {nq} C2 queries cycled for {args.cpu_seconds:.0f} s per mode on the same {args.docs}-doc corpus "
                          f"(posting lists of the sample regenerated on the host); oracle/ss_oracle.c so_search_lex_ref = union_docid_3 "
                          f"sub-query queue over intersection_blockid / single_blockid with block-max pruning, add_topk with docid_hashset",
                "modes": {
                    "throughput_1_shard": {"value": one_tp, "threads": cores, "note": "independent queries per core, one 10 M-doc shard (what one GPU holds)"},
                    "latency_1_shard": {"value": one_lat_qps, "threads": 1, "p50_us": pct(list(one_lat), 50), "p99_us": pct(list(one_lat), 99), "samples": len(one_lat)},
                    "throughput_S_shards": {"value": s_tp, "shards": cores, "threads": cores, "note": "reference default: S = cores document-partitioned "
                                            "shards (index.rs:2055-2062), each query = S shard tasks + merge; every core busy with whole queries"},
                    "latency_S_shards": {"value": s_lat_qps, "shards": cores, "threads": cores, "p50_us": pct(list(s_lat), 50), "p99_us": pct(list(s_lat), 99),
                                         "samples": len(s_lat), "note": "one query at a time, one thread per shard (search.rs:1637-1650)"}},
                "published_reference_point": "BASELINE.md section 2: union mean 439 us on 5 M Wikipedia docs, single thread",
                "seconds": time.perf_counter() - t0}
            del shards

    # ------------------------------------------------------------------ vector (secondary)
    vec = None
    if args.workload in ("all", "vec"):
        kv, B = 100, 64
        t0 = time.perf_counter()
        sh.synth_vectors(O.VEC_SEED, args.rows, args.dim)
        vbuild = time.perf_counter() - t0
        qv_np = O.vec_gen(O.VECQ_SEED, 0, B, args.dim)
        qv = torch.from_numpy(qv_np).to(dev)
        v_doc = torch.empty((B, kv), dtype=torch.int32, device=dev)
        v_score = torch.empty((B, kv), dtype=torch.float32, device=dev)
        v_cnt = torch.empty((B,), dtype=torch.int32, device=dev)
        v_tot = torch.empty((B,), dtype=torch.int64, device=dev)

        def vec_call(n=B):
            N.check(L.ss_vec_search_dev(sh._h, n, qv.data_ptr(), kv, N.FLT_MIN_NEG, v_doc.data_ptr(), v_score.data_ptr(),
                                        v_cnt.data_ptr(), v_tot.data_ptr(), sptr), "ss_vec_search_dev")
            if world > 1:
                merged["vec"] = comm.allgather_merge(v_doc[:n], v_score[:n], v_cnt[:n], kv, sptr)

        vec_call()
        torch.cuda.synchronize()
        sh.profile(True)
        sh.profile_read(1, reset=True)
        vn, dtv = timed_for(vec_call)
        launches, kms = sh.profile_read(1, reset=True)
        sh.profile(False)
        cnt = v_cnt.cpu().numpy()
        assert np.all(cnt.astype(np.int64) == min(kv, args.rows)), "vector candidate overflow or short result in bench"
        flops = 2.0 * args.dim * args.rows * B  # per pass (2*dim*N per query)
        avg_ms = kms / max(launches, 1)
        ach = flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        lat = latencies(vec_call, 400)
        lat1 = latencies(lambda: vec_call(1), 1000)  # <= 32 queries: half of the MFMA work is skipped, the pass is HBM-bound
        hv_doc = np.empty((B, kv), np.uint32); hv_score = np.empty((B, kv), np.float32)
        hv_cnt = np.empty(B, np.uint32); hv_tot = np.empty(B, np.uint64)

        def vec_host_call(n=B):
            N.check(L.ss_vec_search(sh._h, n, N.ptr(qv_np, N.f32p), kv, N.FLT_MIN_NEG, N.ptr(hv_doc, N.u32p), N.ptr(hv_score, N.f32p),
                                    N.ptr(hv_cnt, N.u32p), N.ptr(hv_tot, N.u64p)), "ss_vec_search")
        vec_host_call()
        ve2e = host_latencies(vec_host_call, 100)
        ve2e1 = host_latencies(lambda: vec_host_call(1), 100)
        # a DEEP page of one query (k = 4096 > SS_MAX_K: four passes under the query's exclusion bitmap); its head must be the top-kv list
        vdeep = None
        if not args.quick and world == 1:
            kd = 4096
            dv_doc = np.empty((1, kd), np.uint32); dv_score = np.empty((1, kd), np.float32)
            dv_cnt = np.empty(1, np.uint32); dv_tot = np.empty(1, np.uint64)

            def vec_deep_call():
                N.check(L.ss_vec_search(sh._h, 1, N.ptr(qv_np, N.f32p), kd, N.FLT_MIN_NEG, N.ptr(dv_doc, N.u32p), N.ptr(dv_score, N.f32p),
                                        N.ptr(dv_cnt, N.u32p), N.ptr(dv_tot, N.u64p)), "ss_vec_search")
            vec_host_call(1)
            head_doc, head_score = hv_doc[0].copy(), hv_score[0].copy()
            vec_deep_call()
            nd_ = int(dv_cnt[0])
            assert nd_ == min(kd, args.rows) and np.array_equal(dv_doc[0][:kv], head_doc) and np.array_equal(dv_score[0][:kv], head_score), "deep vector page: head differs"
            assert np.all(dv_score[0][:nd_ - 1] >= dv_score[0][1:nd_]) and len(set(dv_doc[0][:nd_].tolist())) == nd_, "deep vector page: not one descending list of distinct docs"
            vd = host_latencies(vec_deep_call, 10)
            vdeep = {"k": kd, "single_query_ms_p50": pct(vd, 50), "samples": len(vd), "entry_point": "ss_vec_search (host pointers)",
                     "checked": "head == the top-%d call's list bit for bit; descending, distinct docs" % kv}
        vec = dict(qps=B * vn / dtv, ms_per_step=dtv / vn * 1e3, calls=vn, build_s=vbuild, deep_page=vdeep,
                   roofline={"bound": "mfma", "kernel": "vec_scan_kernel (+refine, all row chunks of one pass)", "achieved": ach,
                             "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TF, "traffic": pmc_traffic("vector"),
                             "mfma_util_pmc": pmc_traffic("vector", "mfma_util"),
                             "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": 4.0 * args.dim * args.rows,
                             "hbm_GBs": 4.0 * args.dim * args.rows / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0,
                             "avg_launch_ms": avg_ms, "launches": int(launches)},
                   latency_ms={"batch64_p50": pct(lat, 50), "batch64_p99": pct(lat, 99), "single_query_p50": pct(lat1, 50),
                               "single_query_p99": pct(lat1, 99), "samples": len(lat), "clock": "HIP events on the launch stream"},
                   end_to_end={"value": B / (np.mean(ve2e) * 1e-3), "unit": "queries/s", "entry_point": "ss_vec_search (host pointers, host clock)",
                               "batch64_ms_p50": pct(ve2e, 50), "batch64_ms_p99": pct(ve2e, 99), "single_query_ms_p50": pct(ve2e1, 50),
                               "single_query_ms_p99": pct(ve2e1, 99), "samples": len(ve2e)})
        # ---- full-size parity (C3): streamed oracle scan of the same 10 M x 768 rows (regenerated on the host slice by
        # slice, running TopK), top-100 ids + scores of a sample of the batch
        vs_all = v_score.cpu().numpy().copy()
        ids_all = v_doc.cpu().numpy().copy()
        c3_ref = None
        if rank == 0 and not args.no_parity:
            t0 = time.perf_counter()
            nsv = B
            c3_ref = F.c3_answers(args.rows, args.dim, qv_np[:nsv], kv, part=(rank, world), slice_rows=32768)
            for i in range(nsv):
                F.check_topk(ids_all[i], vs_all[i], c3_ref[i][0], c3_ref[i][1], 1e-4, f"C3 full size f32, query {i}")
            parity["c3_f32"] = {"queries": nsv, "rows": args.rows, "dim": args.dim, "checked": "top-100 rows outside the tie band, scores rtol 1e-4; "
                                "oracle = so_vec_search (dot_f32_avx2 order, TopK::push) streamed over host-regenerated 32 K-row slices",
                                "seconds": time.perf_counter() - t0}

        # ---- AnnMode::Nprobe on the same image: a cluster STRUCTURE (256 records per cluster, 256 clusters per 65 536-doc
        # level, as the reference lays a level out) over the synthetic rows -- the cost of the mode does not depend on what
        # the clusters mean.  16 of 256 clusters per level = 6.25 % of the records per query.
        ann_lc, ann_cc = [], []
        for l0 in range(0, args.rows, 65536):
            n_l = min(65536, args.rows - l0)
            cl = [256] * (n_l // 256) + ([n_l % 256] if n_l % 256 else [])
            ann_lc.append(len(cl)); ann_cc += cl
        ann_mode = S.AnnMode.Nprobe(16)._c()
        v_ncl = torch.empty((B,), dtype=torch.int32, device=dev)

        def ann_leg(i8):
            sh.set_clusters(ann_lc, ann_cc)

            def step(n):
                if i8:
                    N.check(L.ss_vec_search_i8_ann_dev(sh._h, n, q8.data_ptr(), None, kv, N.FLT_MIN_NEG, C.addressof(ann_mode),
                                                       v_doc.data_ptr(), v_score.data_ptr(), v_cnt.data_ptr(), v_tot.data_ptr(),
                                                       v_ncl.data_ptr(), sptr), "ss_vec_search_i8_ann_dev")
                else:
                    N.check(L.ss_vec_search_ann_dev(sh._h, n, qv.data_ptr(), kv, N.FLT_MIN_NEG, C.addressof(ann_mode), v_doc.data_ptr(),
                                                    v_score.data_ptr(), v_cnt.data_ptr(), v_tot.data_ptr(), v_ncl.data_ptr(), sptr),
                            "ss_vec_search_ann_dev")
            l1 = latencies(lambda: step(1), 200)
            l64 = latencies(lambda: step(B), 50)
            assert np.all(v_cnt.cpu().numpy().astype(np.int64) == kv), "ANN candidate overflow or short result in bench"
            return {"mode": "Nprobe(16) of 256 clusters per level", "clusters_visited_per_query": int(v_ncl[0].item()),
                    "single_query_ms_p50": pct(l1, 50), "single_query_ms_p99": pct(l1, 99), "batch64_ms_p50": pct(l64, 50), "samples": len(l1)}

        # ---- C4 hybrid: query i of C2 paired with query i of C3, each side top-100, RRF(0.6), final top-100 --
        # lexical search, vector search and the fusion of the whole batch on the device, nothing crosses PCIe in between
        if bm is not None and world == 1:
            kh = 100
            h_ldoc = torch.empty((B, kh), dtype=torch.int32, device=dev); h_lsc = torch.empty((B, kh), dtype=torch.float32, device=dev)
            h_lcnt = torch.empty((B,), dtype=torch.int32, device=dev); h_ltot = torch.empty((B,), dtype=torch.int64, device=dev)
            h_doc = torch.empty((B, kh), dtype=torch.int64, device=dev); h_sc = torch.empty((B, kh), dtype=torch.float32, device=dev)
            h_src = torch.empty((B, kh), dtype=torch.uint8, device=dev); h_cnt = torch.empty((B,), dtype=torch.int32, device=dev)
            sh.set_strategy(N.BM25_AUTO)

            def hyb_call():
                N.check(L.ss_bm25_search_dev(sh._h, B, q_dev.data_ptr(), kh, N.RT_TOPK, OPS, h_ldoc.data_ptr(), h_lsc.data_ptr(),
                                             h_lcnt.data_ptr(), h_ltot.data_ptr(), sptr), "ss_bm25_search_dev")
                N.check(L.ss_vec_search_dev(sh._h, B, qv.data_ptr(), kv, N.FLT_MIN_NEG, v_doc.data_ptr(), v_score.data_ptr(),
                                            v_cnt.data_ptr(), v_tot.data_ptr(), sptr), "ss_vec_search_dev")
                N.check(L.ss_rrf_merge_dev(local_rank, B, kh, h_ldoc.data_ptr(), h_lcnt.data_ptr(), kv, v_doc.data_ptr(), v_cnt.data_ptr(), 0,
                                           0, kh, h_doc.data_ptr(), h_sc.data_ptr(), h_src.data_ptr(), h_cnt.data_ptr(), sptr),
                        "ss_rrf_merge_dev")
            hyb_call()
            torch.cuda.synchronize()
            # against the host fusion of the same lists (ss_merge_results, the reference's RRF restated on the CPU side)
            for qi in (0, B - 1):
                nl_, nv_ = int(h_lcnt[qi].item()), int(v_cnt[qi].item())
                hd, hs, hsrc = S.merge_results(S.SearchMode.Hybrid,
                                               (h_ldoc[qi, :nl_].cpu().numpy().astype(np.uint64), h_lsc[qi, :nl_].cpu().numpy()),
                                               (v_doc[qi, :nv_].cpu().numpy().astype(np.uint64), v_score[qi, :nv_].cpu().numpy()), 0, kh)
                n_ = int(h_cnt[qi].item())
                assert n_ == len(hd) and np.array_equal(h_doc[qi, :n_].cpu().numpy().astype(np.uint64), hd)
                assert np.array_equal(h_sc[qi, :n_].cpu().numpy(), hs), "device RRF differs from ss_merge_results"
            # full-size parity (C4): the lexical top-100 against the oracle at full size, the vector top-100 (above), and the
            # fusion of the GPU's lists against the oracle's so_merge of the same lists (ranks only enter an RRF score)
            if rank == 0 and not args.no_parity and c3_ref is not None:
                t0 = time.perf_counter()
                nsh = len(c3_ref)
                lans = F.c2_answers_chunked(args.docs, term_lists[:nsh], th, kh, O.OP_OR, O.RT_TOPK, part=(rank, world))
                for i in range(nsh):
                    nl_, nv_ = int(h_lcnt[i].item()), int(v_cnt[i].item())
                    gl = (h_ldoc[i, :nl_].cpu().numpy(), h_lsc[i, :nl_].cpu().numpy())
                    gv = (v_doc[i, :nv_].cpu().numpy(), v_score[i, :nv_].cpu().numpy())
                    F.check_topk(gl[0], gl[1], lans[i][0], lans[i][1], 1e-4, f"C4 full size, lexical top-100, query {i}")
                    F.check_topk(gv[0], gv[1], c3_ref[i][0], c3_ref[i][1], 1e-4, f"C4 full size, vector top-100, query {i}")
                    od, os_, _ = O.merge(2, (gl[0].astype(np.uint64), gl[1]), (gv[0].astype(np.uint64), gv[1]), 0, kh)
                    n_ = int(h_cnt[i].item())
                    assert n_ == len(od) and np.array_equal(h_doc[i, :n_].cpu().numpy().astype(np.uint64), od), f"C4 RRF ids, query {i}"
                    assert np.allclose(h_sc[i, :n_].cpu().numpy(), os_, rtol=1e-6), f"C4 RRF scores, query {i}"
                parity["c4"] = {"queries": nsh, "checked": "BM25 top-100 and cosine top-100 against the full-size oracle lists, device RRF against "
                                "the oracle's so_merge (RRF 0.6) of the same lists", "seconds": time.perf_counter() - t0}
            hn, dth = timed_for(hyb_call)
            vec["hybrid"] = {"workload": "C4: C2 query i + C3 query i, top-100 each, RRF(0.6), final top-100; batch 64, all on device",
                             "value": B * hn / dth, "unit": "queries/s", "ms_per_call": dth / hn * 1e3, "calls": hn}
        # ---- the multi-shard entry points: this rank's shard task + ONE all-gather + merge (hybrid: RRF after the gather)
        # behind the C ABI, host queries in, merged answers out on every rank; with one rank the exchange is the identity.
        # allgather_us = HIP events around the collective alone (ss_comm_profile).
        if bm is not None and comm is not None and not args.quick:
            comm.profile(True)
            sh.set_strategy(N.BM25_AUTO)
            sharded = {}
            qh = np.ascontiguousarray(q_np[:B])

            def leg(name, fn, n_q):
                fn()
                comm.profile_read()
                n_, d_ = timed_for(fn, min_calls=50)
                nc, us = comm.profile_read()
                sharded[name] = {"value": n_q * n_ / d_, "unit": "queries/s", "ms_per_call": d_ / n_ * 1e3, "calls": n_, "queries_per_call": n_q,
                                 "allgather_us": us, "collectives": nc}
            leg("lexical_top10", lambda: comm.search_lexical_sharded(sh, q_np, k, N.RT_TOPK), nq)
            leg("vector_top100", lambda: comm.search_vector_sharded(sh, qv_np, kv), B)
            leg("hybrid_top100", lambda: comm.search_hybrid_sharded(sh, qh, qv_np, 0, 100, N.RT_TOPK), B)
            comm.profile(False)
            sharded["note"] = ("ss_bm25_search_sharded / ss_vec_search_sharded / ss_hybrid_search_sharded: host queries in, this rank's shard "
                               "searched, ONE RCCL all-gather of the lists + totals + status, merge (hybrid: RRF over the cross-shard "
                               f"concatenations) on the device, merged answers on every rank's host; {world} rank(s); value = merged answers per second")
            vec["sharded"] = sharded

        # ---- the reference's REAL calling pattern (search.rs:1637-1743: one search per runtime worker, no batched entry point): T host
        # threads, each issuing ONE query per call through Index::search of the C++ mirror; the C ABI coalesces the concurrent callers
        # into device batches (ss_shard_set_coalescing).  Host clock around every call.
        if bm is not None and world == 1 and rank == 0 and not args.quick and not args.no_concurrent:
            HL = C.CDLL(os.path.join(ROOT, "seekstorm_amd", "lib", "libseekstorm_host.so"))
            HL.ssh_index_adopt.restype = C.c_void_p
            HL.ssh_index_adopt.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
            HL.ssh_index_destroy.argtypes = [C.c_void_p]
            HL.ssh_bench_concurrent.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_double, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
            hs_ = (C.c_void_p * 1)(sh._h)
            hd_ = (C.c_int * 1)(local_rank)
            ixp = HL.ssh_index_adopt(1, hs_, hd_)
            flat = np.array([t for tl in term_lists for t in tl], np.uint32)
            toff = np.zeros(nq + 1, np.uint32)
            toff[1:] = np.cumsum([len(tl) for tl in term_lists])
            conc = {}
            for name, mode, T, nqq, length in (("lexical_top10_T64", N.MODE_LEXICAL, 64, nq, 10), ("lexical_top10_T256", N.MODE_LEXICAL, 256, nq, 10),
                                               ("vector_top100_T64", N.MODE_VECTOR, 64, B, 100), ("vector_top100_T256", N.MODE_VECTOR, 256, B, 100),
                                               ("hybrid_top100_T64", N.MODE_HYBRID, 64, B, 100), ("hybrid_top100_T256", N.MODE_HYBRID, 256, B, 100)):
                out5 = (C.c_double * 5)()
                st0 = sh.coalescing_stats()
                # (256 callers of a 9 ms pass make ~27 calls a second each: twice the time, so that one stalled pass -- 256 calls late at once -- is
                # not by itself the leg's p99; profiles/r6g_tail_v256.log: three isolated runs p99 / p50 1.01, 1.01 and, with ONE pass of 18 ms, 1.29)
                leg_s = max(args.min_seconds, 1.0) * (2.0 if (T >= 256 and mode != N.MODE_LEXICAL) else 1.0)
                N.check(HL.ssh_bench_concurrent(ixp, mode, T, leg_s, nqq, flat.ctypes.data, toff.ctypes.data, qv_np.ctypes.data,
                                                int(S.QueryType.Union), length, N.RT_TOPK, out5), "ssh_bench_concurrent")
                st1 = sh.coalescing_stats()
                lb, lq, vb, vq = (st1[i] - st0[i] for i in range(4))
                conc[name] = {"value": out5[0] / out5[1], "unit": "queries/s", "threads": T, "searches": int(out5[0]), "seconds": out5[1],
                              "latency_us_p50": out5[2], "latency_us_p99": out5[3], "errors": int(out5[4]),
                              "mean_lexical_batch": (lq / lb) if lb else None, "mean_vector_batch": (vq / vb) if vb else None}
                assert out5[4] == 0, f"concurrent leg {name}: {int(out5[4])} searches failed"
            HL.ssh_index_destroy(ixp)
            conc["note"] = ("T host threads, each ONE query per call through Index::search (C++ mirror) -> ss_bm25_search / ss_vec_search; the C ABI "
                            "merges the concurrent callers into device batches (group commit: whoever arrives while a batch runs joins the next "
                            "one); host clock per call; mean_*_batch = queries per merged device batch")
            vec["concurrent_callers"] = conc
        if args.rows >= 65536:
            vec["ann"] = ann_leg(False)
        # property checks at full size: sorted, and the scores really are dot products of the returned rows
        vec_call()
        torch.cuda.synchronize()
        vs = v_score.cpu().numpy()
        assert np.all(vs[:, :-1] >= vs[:, 1:])
        ids = v_doc.cpu().numpy()
        r0 = sh.read_rows(int(ids[0, 0]), 1)[0]
        assert abs(float(r0 @ qv_np[0]) - float(vs[0, 0])) < 1e-4
        if rank == 0 and world == 1 and not args.no_cpu:
            from concurrent.futures import ThreadPoolExecutor
            # cpu_baseline: AnnMode::All scans in the reference's structure.  (a) the reference's default execution -- S = cores
            # document-partitioned shards, one task per shard and query (index.rs:2055-2062, search.rs:1637-1650) -- on the FULL matrix
            # (10 M x 768 regenerated on the host, 30.7 GB) when the host has the memory: measured, nothing extrapolated;
            # (b) throughput mode (every core answers whole queries) on a 1 M-row sample, scaled linearly, beside it.
            eff_cpus, cpu_info = F.effective_cpus()
            cores = max(1, min(F.host_threads(128), int(np.ceil(eff_cpus))))  # threads spawned: one per CPU the process can really run on
            try:
                avail_gb = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 1e6
            except Exception:
                avail_gb = 0.0
            full = avail_gb > 4.0 * args.dim * args.rows / 1e9 * 1.5
            M = args.rows if full else min(args.rows, 1_000_000)
            rows = np.empty((M, args.dim), np.float32)
            cuts = np.linspace(0, M, cores * 8 + 1).astype(np.int64)

            def gen_slice(i):
                rows[cuts[i]:cuts[i + 1]] = O.vec_gen(O.VEC_SEED, int(cuts[i]), int(cuts[i + 1] - cuts[i]), args.dim)
            t0 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                list(ex.map(gen_slice, range(cores * 8)))
            gen_s = time.perf_counter() - t0
            scale = args.rows / M
            qps_l, done_l, lat = O.bench_vec(rows, qv_np, kv, 1, cores, args.cpu_seconds)
            Ms = min(M, 1_000_000)
            qps_t, done_t, _ = O.bench_vec(rows[:Ms], qv_np, kv, 0, cores, args.cpu_seconds)
            vec["cpu_baseline"] = {
                "value": qps_l / scale, "unit": "queries/s", "cores": eff_cpus, "threads": cores, "cpu_info": cpu_info, "kind": "port", "rows": int(M),
                "host_generation_s": gen_s,
                "sample": (f"{done_l} top-100 queries over {'ALL' if full else 'the first'} {M} rows in {args.cpu_seconds:.0f}s: one query at a time, the rows "
                           f"split over {cores} workers + merge = the reference's default S = cores shards, one task per shard and query "
                           f"(oracle so_bench_vec: dot_f32_avx2 order + TopK::push)" + ("" if full else f"; rate divided by {scale:.0f} (linear scan) to {args.rows} rows")),
                "latency_p50_ms": float(np.percentile(lat, 50)) * scale / 1e3 if len(lat) else None, "latency_samples": int(len(lat)),
                "throughput_mode_sample": {"queries_per_s": qps_t / (args.rows / Ms), "rows": int(Ms),
                                           "sample": f"{done_t} scans of a {Ms}-row slice, {cores} threads each answering whole queries; rate divided by "
                                                     f"{args.rows / Ms:.0f} to {args.rows} rows"}}
            del rows

        # ---- the same corpus as Precision::I8 records (quantize_f32_to_i8 of the same rows): HBM-bound stream kernel
        t0 = time.perf_counter()
        sh.synth_vectors_i8(O.VEC_SEED, args.rows, args.dim)
        build8 = time.perf_counter() - t0
        q8 = torch.from_numpy(O.quantize_i8(qv_np)).to(dev)

        def vec8_call():
            N.check(L.ss_vec_search_i8_dev(sh._h, B, q8.data_ptr(), None, kv, N.FLT_MIN_NEG, v_doc.data_ptr(), v_score.data_ptr(),
                                           v_cnt.data_ptr(), v_tot.data_ptr(), sptr), "ss_vec_search_i8_dev")
        vec8_call()
        torch.cuda.synchronize()
        sh.profile(True)
        sh.profile_read(1, reset=True)
        n8, dt8 = timed_for(vec8_call)
        launches8, kms8 = sh.profile_read(1, reset=True)
        sh.profile(False)
        assert np.all(v_cnt.cpu().numpy().astype(np.int64) == min(kv, args.rows))
        vs8, id8 = v_score.cpu().numpy(), v_doc.cpu().numpy()
        assert np.all(vs8[:, :-1] >= vs8[:, 1:])
        r8 = sh.read_rows_i8(int(id8[0, 0]), 1)[0]
        assert float(r8.astype(np.int64) @ q8[0].cpu().numpy().astype(np.int64)) == float(vs8[0, 0]), "i8 score is not the integer dot product"
        if rank == 0 and not args.no_parity:
            t0 = time.perf_counter()
            nsv = B
            ref8 = F.c3_answers(args.rows, args.dim, qv_np[:nsv], kv, part=(rank, world), slice_rows=32768, i8=True)
            for i in range(nsv):
                assert np.array_equal(vs8[i], ref8[i][1]), f"C3 full size i8: scores of query {i} differ from the oracle (integer dot products: ==)"
                F.check_topk(id8[i], vs8[i], ref8[i][0], ref8[i][1], 0.0, f"C3 full size i8, query {i}")
            parity["c3_i8"] = {"queries": nsv, "checked": "top-100 scores == (integer dots), rows outside the k-th score's tie group",
                               "seconds": time.perf_counter() - t0}
        ms8 = kms8 / max(launches8, 1)
        bytes8 = 1.0 * args.dim * args.rows
        gbs8 = bytes8 / (ms8 * 1e-3) / 1e9 if ms8 > 0 else 0.0
        ann8 = ann_leg(True) if args.rows >= 65536 else None
        vec["i8"] = {"metric": "queries/sec (i8 dot top-100, batch 64)", "ann": ann8, "value": B * n8 / dt8, "ms_per_call": dt8 / n8 * 1e3,
                     "calls": n8, "build_s": build8,
                     "roofline": {"bound": "hbm", "kernel": "vec8_scan_kernel (+refine, all row chunks of one pass)", "achieved": gbs8,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs8 / HBM_PEAK_GBS, "traffic": pmc_traffic("vector_i8"),
                                  "algorithmic_bytes_per_launch": bytes8, "algorithmic_ops_per_launch": 2.0 * args.dim * args.rows * B,
                                  "avg_launch_ms": ms8, "launches": int(launches8)}}

    # ------------------------------------------------------------------ report
    if rank == 0:
        prim = bm if bm is not None else vec
        is_bm = bm is not None
        gen_note = ("synthetic generators follow SURVEY 8d, integer-exact so that host and device generate identical data: tf = 1 + "
                    "geometric(p = 0.6) drawn from 32 hash bits by thresholds floor(0.4^m 2^32); vector components uniform(-1, 1) "
                    "then normalize_f32 instead of Box-Muller (f32 log / cos differ between host libm and the device; after "
                    "normalisation to the unit sphere at dim 768 both give dot products ~ N(0, 1/768), and a brute-force scan does the "
                    "same work whatever the values)")
        line = {
            "metric": "queries/sec" + (" (BM25 3-term OR top-10)" if is_bm else " (cosine top-100, batch 64)"),
            "value": prim["qps"],
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps if is_bm else prim["calls"],
            "warmup": args.warmup,
            "ms_per_step": prim["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 sums of 16-bit posting weight codes (BM25); f32 MFMA (vector)" if is_bm else "f32",
            "data": "synthetic",
            "config": ({"workload": "C2: 10M synthetic docs, 3-term OR BM25 top-10, one shard per GPU; host pointers in, answers on the host", "docs_per_shard": args.docs,
                        "queries_per_call": args.queries, "calls_per_step": args.calls_per_step, "k": 10, "vocabulary": 4096,
                        "result_type": "Topk", "strategy": "auto (pruned top-k; exhaustive scan reported beside it)",
                        "corpus": f"shard r of {world} of ONE generator stream of {args.docs * world} docs (doc g -> shard g % {world})",
                        "value_counts": "merged answers per second over all shards (every shard sees every query; weak scaling: the corpus "
                                        "grows with N, so a flat value is perfect scaling)", "generators": gen_note} if is_bm else
                       {"workload": "C3: 10M x 768 f32, batch-64 cosine top-100 brute force", "rows_per_shard": args.rows,
                        "dim": args.dim, "batch": 64, "k": 100, "generators": gen_note}),
            "global_qps": prim["qps"],
            "shard_queries_per_s": prim["qps"] * world,
            "docs_scanned_per_s": prim["qps"] * world * (args.docs if is_bm else args.rows),
            "roofline": prim["roofline"],
            "cpu_baseline": prim.get("cpu_baseline"),
            "latency_ms": prim["latency_ms"],
            "end_to_end": prim.get("end_to_end"),
            "value_protocol": ("ss_bm25_search: host pointers in, answers on the host (SURVEY 8d: submit -> results on host)" if is_bm else
                               "ss_vec_search_dev (device resident)"),
            "device_resident": prim.get("device_resident"),
            "parity_full_size": parity or None,
        }
        if is_bm:
            line["exhaustive"] = bm["exhaustive"]
            line["topk_count"] = bm["topk_count"]
            if bm.get("intersection"):
                line["intersection"] = bm["intersection"]
            if bm.get("exhaustive_not_tombstones"):
                line["exhaustive_not_tombstones"] = bm["exhaustive_not_tombstones"]
            if bm.get("clustered"):
                line["clustered"] = bm["clustered"]
            if bm.get("union16"):
                line["union16"] = bm["union16"]
            if bm.get("deep_page"):
                line["deep_page"] = bm["deep_page"]
            if bm.get("scale_check"):
                line["scale_check"] = bm["scale_check"]
            if "rationed_vocabulary" in bm:
                line["rationed_vocabulary"] = bm["rationed_vocabulary"]
            if "multi_field" in bm:
                line["multi_field"] = bm["multi_field"]
            if "realistic_vocabulary" in bm:
                line["realistic_vocabulary"] = bm["realistic_vocabulary"]
            if "real_format" in bm:
                line["real_format"] = bm["real_format"]
            if "commit" in bm:
                line["commit"] = bm["commit"]
            line["bm25"] = {"build_s": bm["build_s"], "postings": int(bm["info"]["n_postings"]), "avgdl": bm["info"]["avgdl"],
                            "mean_algorithmic_bytes_per_query": bm["mean_bytes_per_query"], "mean_union_size": bm["mean_union"]}
        if vec is not None and is_bm:
            line["vector"] = {"metric": "queries/sec (cosine top-100, 10M x 768 f32, batch 64)", "value": vec["qps"],
                              "global_qps": vec["qps"], "ms_per_call": vec["ms_per_step"], "calls": vec["calls"], "roofline": vec["roofline"],
                              "cpu_baseline": vec.get("cpu_baseline"), "latency_ms": vec["latency_ms"], "end_to_end": vec.get("end_to_end"),
                              "build_s": vec["build_s"], "rows_per_shard": args.rows, "dim": args.dim, "hybrid": vec.get("hybrid"),
                              "ann": vec.get("ann"), "i8": vec.get("i8"), "deep_page": vec.get("deep_page")}
            if vec.get("sharded"):
                line["sharded"] = vec["sharded"]
            if vec.get("concurrent_callers"):
                line["concurrent_callers"] = vec["concurrent_callers"]
        elif vec is not None:
            line["i8"] = vec.get("i8")
        final_line = line
    else:
        final_line = None
    if comm is not None:
        comm.close()
    sh.close()
    if world > 1:
        dist.destroy_process_group()
    if final_line is not None:  # LAST: whatever the libraries print while they shut down stands before the line the driver parses
        emit(final_line, world)


if __name__ == "__main__":
    main()
