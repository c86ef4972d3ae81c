#!/usr/bin/env python3
"""bench.py -- SeekStorm query hot path on MI355X.

Primary workload (BASELINE.json configs[1], "C2"): 10M synthetic docs, 3-term OR, BM25 top-10, one shard per GPU.
A "step" = one batch of 1000 resolved queries through ss_bm25_search_dev (queries and outputs resident in HBM).
Secondary workload (configs[2], "C3"), reported in the same JSON line under "vector": 10M x 768 f32, batch-64 cosine
top-100 brute force through ss_vec_search_dev.

Multi-GPU (weak scaling): one process per GPU, one 10M-doc / 10M-row shard per rank, every rank answers
the same query batch on its shard, one RCCL all-gather of the per-shard top-k, ss_topk_merge_dev on every rank.
`value` counts shard-queries (queries x shards scanned) per second so that it is the whole-job aggregate;
`global_qps` is the rate of merged answers over the N-times larger corpus.

  python bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: FP32 matrix peak


def pmc_traffic(kind):
    """HBM bytes per launch from the committed PMC run (profiles/pmc_traffic.json, produced by tools/pmc_summary.py
    from separate rocprofv3 --pmc passes of this same command); None if the file is absent."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[kind]["hbm_bytes_per_launch"]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="all", choices=["all", "bm25", "vec"])
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-topk-count", action="store_true", help="skip the TopkCount comparison (keeps counter profiles of the Topk kernels clean)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

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
    L = S.lib()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # an explicit (non-null) stream: the C ABI treats a NULL stream as "the shard's own stream", and the RCCL
    # all-gather must be ordered after the search kernels on ONE stream
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    sptr = C.c_void_p(stream.cuda_stream)
    assert sptr.value, "expected a non-null HIP stream handle"
    sh = S.Shard(local_rank, shard_id=rank)
    from seekstorm_amd import distributed as D
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

    def latencies(step_fn, n):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in ev:
            a.record(stream)
            step_fn()
            b.record(stream)
            torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in ev]

    # ------------------------------------------------------------------ BM25 (primary)
    bm = None
    if args.workload in ("all", "bm25"):
        from oracle import oracle as O  # query set + thresholds come from the generator module (host constants only)
        k = 10
        term_lists, th = make_c2_queries(O, args.queries)
        tab = O.len_table()
        t0 = time.perf_counter()
        sh.synth_lexical(O.LEX_SEED ^ (rank * 0x9E3779B1), args.docs, th, tab)
        build_s = time.perf_counter() - t0
        info = sh.lexical_info()
        q_np = sh.make_queries(term_lists, S.QueryType.Union)
        nq = len(q_np)
        q_dev = torch.from_numpy(q_np.view(np.uint8).reshape(nq, -1).copy()).to(dev)
        o_doc = torch.empty((nq, k), dtype=torch.int32, device=dev)
        o_score = torch.empty((nq, k), dtype=torch.float32, device=dev)
        o_cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
        o_tot = torch.empty((nq,), dtype=torch.int64, device=dev)
        def bm_step(n=nq):
            N.check(L.ss_bm25_search_dev(sh._h, n, q_dev.data_ptr(), k, N.RT_TOPK, 2 | (3 << 8), o_doc.data_ptr(), o_score.data_ptr(),
                                         o_cnt.data_ptr(), o_tot.data_ptr(), sptr), "ss_bm25_search_dev")
            if world > 1:  # one all-gather of the per-shard top-k over RCCL, identical merge on every rank
                g = D.all_gather_topk_packed(o_doc[:n], o_score[:n], o_cnt[:n])  # ONE collective per batch
                merged["bm25"] = D.merge_gathered_device_packed(g, n, k, sptr, local_rank)

        # exact union sizes for the roofline's "1 B per scored candidate" term: one untimed TopkCount pass (exhaustive scan)
        sh.set_strategy(N.BM25_EXHAUSTIVE)
        N.check(L.ss_bm25_search_dev(sh._h, nq, q_dev.data_ptr(), k, N.RT_TOPKCOUNT, 2 | (3 << 8), o_doc.data_ptr(),
                                     o_score.data_ptr(), o_cnt.data_ptr(), o_tot.data_ptr(), sptr), "ss_bm25_search_dev")
        torch.cuda.synchronize()
        tot = o_tot.cpu().numpy().astype(np.int64)
        ref_scores = o_score.cpu().numpy().copy()
        # algorithmic bytes: sum_t df_t*(2B id + 1B tf) + 1B per scored candidate + 4B per (term, block) + 8B*k
        uniq = sorted({t for tl in term_lists for t in tl})
        dfm = dict(zip(uniq, (int(x) for x in sh.posting_count(uniq))))
        n_blocks = (args.docs + 65535) // 65536
        bytes_q = np.array([sum(dfm[t] for t in tl) * 3 + int(tot[i]) + 4 * n_blocks * len(tl) + 8 * k
                            for i, tl in enumerate(term_lists)], np.float64)
        bytes_launch = float(bytes_q.sum())

        def measure(strategy, steps, warmup):
            """timed run of one strategy: (qps, ms_per_step, avg kernel ms from the library's HIP events, launches)"""
            sh.set_strategy(strategy)
            sh.profile(True)
            bm_step()
            torch.cuda.synchronize()
            assert np.array_equal(ref_scores, o_score.cpu().numpy()), "Topk ranking differs from the exhaustive TopkCount pass"
            sh.profile_read(0, reset=True)
            dt_ = timed(bm_step, steps, warmup)
            launches_, kms_ = sh.profile_read(0, reset=True)
            sh.profile(False)
            return nq * steps / dt_, dt_ / steps * 1e3, kms_ / max(launches_, 1), int(launches_)

        # (1) exhaustive scan: every posting of every query term is read -- the kernel the HBM roofline is about
        ex_qps, ex_ms, ex_kms, ex_n = measure(N.BM25_EXHAUSTIVE, max(4, args.steps // 2), min(args.warmup, 2))
        # (2) the default strategy (AUTO): top-k unions take the pruned path (MaxScore over the probe index)
        qps, ms_step, avg_ms, launches = measure(N.BM25_AUTO, args.steps, args.warmup)
        dt = nq * args.steps / qps

        # (3) ResultType::TopkCount, the reference server's default: pruned top-k + exact union counts (popcounts over the
        # probe index's bit records) under AUTO, against the exhaustive scan that used to serve it
        def tc_step():
            N.check(L.ss_bm25_search_dev(sh._h, nq, q_dev.data_ptr(), k, N.RT_TOPKCOUNT, 2 | (3 << 8), o_doc.data_ptr(),
                                         o_score.data_ptr(), o_cnt.data_ptr(), o_tot.data_ptr(), sptr), "ss_bm25_search_dev")
        tc = {}
        for name, strat in (() if args.no_topk_count else (("exhaustive", N.BM25_EXHAUSTIVE), ("auto", N.BM25_AUTO))):
            sh.set_strategy(strat)
            tc_step()
            torch.cuda.synchronize()
            assert np.array_equal(tot, o_tot.cpu().numpy().astype(np.int64)), "TopkCount totals differ between strategies"
            assert np.array_equal(ref_scores, o_score.cpu().numpy())
            d_ = timed(tc_step, max(4, args.steps // 2), min(args.warmup, 2))
            tc[name] = {"value": nq * max(4, args.steps // 2) / d_, "unit": "queries/s", "ms_per_step": d_ / max(4, args.steps // 2) * 1e3}
        sh.set_strategy(N.BM25_AUTO)
        ach = bytes_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        ex_ach = bytes_launch / (ex_kms * 1e-3) / 1e9 if ex_kms > 0 else 0.0
        lat_batch = latencies(bm_step, 12)
        lat_one = latencies(lambda: bm_step(1), 60)
        # Roofline of the dominant kernel of the timed region (bm25_probe_kernel).  A pruning kernel answers WITHOUT reading
        # most of SURVEY 8d's algorithmic bytes (all postings of the query terms), so dividing those by its time gives a rate
        # above the HBM peak that says nothing about the kernel.  `achieved` is therefore what it really moves per launch
        # (PMC FETCH_SIZE / WRITE_SIZE of the same command, profiles/pmc_traffic.json) over the live launch time; the rate on
        # the algorithmic bytes is kept beside it, and the exhaustive scan -- the kernel that does stream them -- below.
        moved = pmc_traffic("bm25_pruned")
        real = (moved if moved else bytes_launch) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        bm = dict(qps=qps, ms_per_step=ms_step, build_s=build_s, info=info,
                  roofline={"bound": "hbm", "kernel": "bm25_probe_kernel<3,1> (pruned strategy: essential terms' postings + probe records)",
                            "achieved": real, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": real / HBM_PEAK_GBS,
                            "traffic": moved, "algorithmic_bytes_per_launch": bytes_launch,
                            "effective_GBs_on_algorithmic_bytes": ach,
                            "avg_launch_ms": avg_ms, "launches": launches,
                            "note": "achieved = bytes the kernel really moves (PMC, committed profile) / live kernel time; it answers "
                                    "without reading most of the SURVEY 8d algorithmic bytes, hence effective_GBs_on_algorithmic_bytes "
                                    "can exceed the peak; the exhaustive scan below is the kernel that streams the algorithmic bytes"},
                  exhaustive={"value": ex_qps, "unit": "queries/s", "ms_per_step": ex_ms,
                              "roofline": {"bound": "hbm", "kernel": "bm25_scan_fast_kernel<3,false,1>", "achieved": ex_ach,
                                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ex_ach / HBM_PEAK_GBS,
                                           "traffic": pmc_traffic("bm25"), "algorithmic_bytes_per_launch": bytes_launch,
                                           "avg_launch_ms": ex_kms, "launches": ex_n}},
                  topk_count=dict(tc, note="same batch with ResultType::TopkCount (exact result_count_total, the reference server's "
                                           "default): AUTO = pruned top-k + union counts from the probe index's bit records"),
                  latency_ms={"batch_p50": pct(lat_batch, 50), "batch_p99": pct(lat_batch, 99),
                              "single_query_p50": pct(lat_one, 50), "single_query_p99": pct(lat_one, 99)},
                  mean_bytes_per_query=float(bytes_q.mean()), mean_union=float(tot.mean()))
        # correctness guard inside the bench: sorted, k results, counts sane
        sc = o_score.cpu().numpy()
        assert np.all(sc[:, :-1] >= sc[:, 1:]) and np.all(o_cnt.cpu().numpy() == k)

        if rank == 0 and world == 1 and not args.no_cpu:
            # cpu_baseline: the oracle (reference-structured C port: containers, block-max pruned table scan) on the
            # host cores, bounded sample of the same query set on the same corpus (terms regenerated on the host)
            from concurrent.futures import ThreadPoolExecutor
            ns = 12
            sample = term_lists[:ns]
            voc = sorted({t for tl in sample for t in tl})
            dl = O.lex_doclen(args.docs, O.LEX_SEED)
            offs, docs, tfs = O.lex_corpus(args.docs, voc, O.LEX_SEED)
            osh = O.Shard(args.docs, dl, offs, docs, tfs)
            remap = {t: i for i, t in enumerate(voc)}
            qs = [[remap[t] for t in tl] for tl in sample]
            cores = min(os.cpu_count() or 1, 32)
            od, os_, _ = osh.search(qs[0], O.OP_OR, k, O.RT_TOPK)  # parity spot check GPU vs oracle on query 0
            assert np.allclose(os_, sc[0][:len(os_)], rtol=1e-4), "bench parity spot check failed"
            done = 0
            t0 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                while time.perf_counter() - t0 < args.cpu_seconds:
                    list(ex.map(lambda q: osh.search(q, O.OP_OR, k, O.RT_TOPK), qs * max(1, cores // ns + 1)))
                    done += len(qs) * max(1, cores // ns + 1)
            el = time.perf_counter() - t0
            bm["cpu_baseline"] = {"value": done / el, "unit": "queries/s", "cores": cores, "kind": "port",
                                  "sample": f"{ns} of the {nq} C2 queries repeated for {el:.1f}s on the same {args.docs}-doc "
                                            f"corpus, {cores} threads, oracle/ss_oracle.c so_search_lex (OR, Topk, k=10)"}
            del osh

    # ------------------------------------------------------------------ vector (secondary)
    vec = None
    if args.workload in ("all", "vec"):
        from oracle import oracle as O
        kv, B = 100, 64
        t0 = time.perf_counter()
        sh.synth_vectors(O.VEC_SEED ^ (rank * 0x9E3779B1), args.rows, args.dim)
        vbuild = time.perf_counter() - t0
        qv = torch.from_numpy(O.vec_gen(O.VECQ_SEED, 0, B, args.dim)).to(dev)
        v_doc = torch.empty((B, kv), dtype=torch.int32, device=dev)
        v_score = torch.empty((B, kv), dtype=torch.float32, device=dev)
        v_cnt = torch.empty((B,), dtype=torch.int32, device=dev)
        v_tot = torch.empty((B,), dtype=torch.int64, device=dev)
        def vec_step(n=B):
            N.check(L.ss_vec_search_dev(sh._h, n, qv.data_ptr(), kv, N.FLT_MIN_NEG, v_doc.data_ptr(), v_score.data_ptr(),
                                        v_cnt.data_ptr(), v_tot.data_ptr(), sptr), "ss_vec_search_dev")
            if world > 1:
                g = D.all_gather_topk_packed(v_doc[:n], v_score[:n], v_cnt[:n])
                merged["vec"] = D.merge_gathered_device_packed(g, n, kv, sptr, local_rank)

        sh.profile(True)
        vec_step()
        torch.cuda.synchronize()
        sh.profile_read(1, reset=True)
        vsteps = max(4, args.steps // 2)
        dtv = timed(vec_step, vsteps, min(args.warmup, 2))
        launches, kms = sh.profile_read(1, reset=True)
        sh.profile(False)
        cnt = v_cnt.cpu().numpy()
        assert np.all(cnt.astype(np.int64) == min(kv, args.rows)), "vector candidate overflow or short result in bench"
        flops = 2.0 * args.dim * args.rows * B  # per pass (2*dim*N per query)
        avg_ms = kms / max(launches, 1)
        ach = flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        lat = latencies(vec_step, 8)
        lat1 = latencies(lambda: vec_step(1), 8)  # <= 32 queries: half of the MFMA work is skipped, the pass is HBM-bound
        vec = dict(qps=B * vsteps / dtv, ms_per_step=dtv / vsteps * 1e3, build_s=vbuild,
                   roofline={"bound": "mfma", "kernel": "vec_scan_kernel (+refine, all row chunks of one pass)", "achieved": ach,
                             "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TF, "traffic": pmc_traffic("vector"),
                             "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": 4.0 * args.dim * args.rows,
                             "hbm_GBs": 4.0 * args.dim * args.rows / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0,
                             "avg_launch_ms": avg_ms, "launches": int(launches)},
                   latency_ms={"batch64_p50": pct(lat, 50), "batch64_p99": pct(lat, 99), "single_query_p50": pct(lat1, 50),
                               "single_query_p99": pct(lat1, 99)})
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
            l1 = latencies(lambda: step(1), 8)
            l64 = latencies(lambda: step(B), 4)
            assert np.all(v_cnt.cpu().numpy().astype(np.int64) == kv), "ANN candidate overflow or short result in bench"
            return {"mode": "Nprobe(16) of 256 clusters per level", "clusters_visited_per_query": int(v_ncl[0].item()),
                    "single_query_ms_p50": pct(l1, 50), "batch64_ms_p50": pct(l64, 50)}
        # ---- C4 hybrid: query i of C2 paired with query i of C3, each side top-100, RRF(0.6), final top-100 --
        # lexical search, vector search and the fusion of the whole batch on the device, nothing crosses PCIe in between
        if bm is not None and world == 1:
            kh = 100
            h_ldoc = torch.empty((B, kh), dtype=torch.int32, device=dev); h_lsc = torch.empty((B, kh), dtype=torch.float32, device=dev)
            h_lcnt = torch.empty((B,), dtype=torch.int32, device=dev); h_ltot = torch.empty((B,), dtype=torch.int64, device=dev)
            h_doc = torch.empty((B, kh), dtype=torch.int64, device=dev); h_sc = torch.empty((B, kh), dtype=torch.float32, device=dev)
            h_src = torch.empty((B, kh), dtype=torch.uint8, device=dev); h_cnt = torch.empty((B,), dtype=torch.int32, device=dev)
            sh.set_strategy(N.BM25_AUTO)
            def hyb_step():
                N.check(L.ss_bm25_search_dev(sh._h, B, q_dev.data_ptr(), kh, N.RT_TOPK, 2 | (3 << 8), h_ldoc.data_ptr(), h_lsc.data_ptr(),
                                             h_lcnt.data_ptr(), h_ltot.data_ptr(), sptr), "ss_bm25_search_dev")
                N.check(L.ss_vec_search_dev(sh._h, B, qv.data_ptr(), kv, N.FLT_MIN_NEG, v_doc.data_ptr(), v_score.data_ptr(),
                                            v_cnt.data_ptr(), v_tot.data_ptr(), sptr), "ss_vec_search_dev")
                N.check(L.ss_rrf_merge_dev(local_rank, B, kh, h_ldoc.data_ptr(), h_lcnt.data_ptr(), kv, v_doc.data_ptr(), v_cnt.data_ptr(), 0,
                                           0, kh, h_doc.data_ptr(), h_sc.data_ptr(), h_src.data_ptr(), h_cnt.data_ptr(), sptr),
                        "ss_rrf_merge_dev")
            hyb_step()
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
            hsteps = max(4, args.steps // 4)
            dth = timed(hyb_step, hsteps, 1)
            vec["hybrid"] = {"workload": "C4: C2 query i + C3 query i, top-100 each, RRF(0.6), final top-100; batch 64, all on device",
                             "value": B * hsteps / dth, "unit": "queries/s", "ms_per_step": dth / hsteps * 1e3}
        if args.rows >= 65536:
            vec["ann"] = ann_leg(False)
        # property checks at full size: sorted, and the scores really are dot products of the returned rows
        vec_step()
        torch.cuda.synchronize()
        vs = v_score.cpu().numpy()
        assert np.all(vs[:, :-1] >= vs[:, 1:])
        ids = v_doc.cpu().numpy()
        r0 = sh.read_rows(int(ids[0, 0]), 1)[0]
        assert abs(float(r0 @ qv[0].cpu().numpy()) - float(vs[0, 0])) < 1e-4
        if rank == 0 and world == 1 and not args.no_cpu:
            from concurrent.futures import ThreadPoolExecutor
            M = min(args.rows, 200_000)
            rows = O.vec_gen(O.VEC_SEED, 0, M, args.dim)
            qh = qv.cpu().numpy()
            cores = min(os.cpu_count() or 1, 32)
            od, os_, _, _ = O.vec_search(rows, qh[0], kv)
            done = 0
            t0 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                while time.perf_counter() - t0 < args.cpu_seconds:
                    list(ex.map(lambda q: O.vec_search(rows, q, kv), [qh[i % B] for i in range(cores)]))
                    done += cores
            el = time.perf_counter() - t0
            scale = args.rows / M
            vec["cpu_baseline"] = {"value": done / el / scale, "unit": "queries/s", "cores": cores, "kind": "port",
                                   "sample": f"{done} cosine top-100 scans of the first {M} rows in {el:.1f}s, {cores} threads, "
                                             f"oracle so_vec_search (dot_f32_avx2 order, TopK::push); rate divided by {scale:.0f} "
                                             f"(linear scan) to {args.rows} rows"}

        # ---- the same corpus as Precision::I8 records (quantize_f32_to_i8 of the same rows): HBM-bound stream kernel
        t0 = time.perf_counter()
        sh.synth_vectors_i8(O.VEC_SEED ^ (rank * 0x9E3779B1), args.rows, args.dim)
        build8 = time.perf_counter() - t0
        q8 = torch.from_numpy(O.quantize_i8(O.vec_gen(O.VECQ_SEED, 0, B, args.dim))).to(dev)
        def vec8_step():
            N.check(L.ss_vec_search_i8_dev(sh._h, B, q8.data_ptr(), None, kv, N.FLT_MIN_NEG, v_doc.data_ptr(), v_score.data_ptr(),
                                           v_cnt.data_ptr(), v_tot.data_ptr(), sptr), "ss_vec_search_i8_dev")
        sh.profile(True)
        vec8_step()
        torch.cuda.synchronize()
        sh.profile_read(1, reset=True)
        dt8 = timed(vec8_step, vsteps * 2, min(args.warmup, 2))
        launches8, kms8 = sh.profile_read(1, reset=True)
        sh.profile(False)
        assert np.all(v_cnt.cpu().numpy().astype(np.int64) == min(kv, args.rows))
        vs8, id8 = v_score.cpu().numpy(), v_doc.cpu().numpy()
        assert np.all(vs8[:, :-1] >= vs8[:, 1:])
        r8 = sh.read_rows_i8(int(id8[0, 0]), 1)[0]
        assert float(r8.astype(np.int64) @ q8[0].cpu().numpy().astype(np.int64)) == float(vs8[0, 0]), "i8 score is not the integer dot product"
        ms8 = kms8 / max(launches8, 1)
        bytes8 = 1.0 * args.dim * args.rows
        gbs8 = bytes8 / (ms8 * 1e-3) / 1e9 if ms8 > 0 else 0.0
        ann8 = ann_leg(True) if args.rows >= 65536 else None
        vec["i8"] = {"metric": "queries/sec (i8 dot top-100, batch 64)", "ann": ann8, "value": B * vsteps * 2 / dt8 * world, "ms_per_step": dt8 / (vsteps * 2) * 1e3,
                     "build_s": build8,
                     "roofline": {"bound": "hbm", "kernel": "vec8_scan_kernel (+refine, all row chunks of one pass)", "achieved": gbs8,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs8 / HBM_PEAK_GBS, "traffic": pmc_traffic("vector_i8"),
                                  "algorithmic_bytes_per_launch": bytes8, "algorithmic_ops_per_launch": 2.0 * args.dim * args.rows * B,
                                  "avg_launch_ms": ms8, "launches": int(launches8)}}

    # ------------------------------------------------------------------ report
    if rank == 0:
        prim = bm if bm is not None else vec
        is_bm = bm is not None
        line = {
            "metric": "queries/sec" + (" (BM25 3-term OR top-10)" if is_bm else " (cosine top-100, batch 64)"),
            "value": prim["qps"] * world,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps if is_bm else max(4, args.steps // 2),
            "warmup": args.warmup,
            "ms_per_step": prim["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": ({"workload": "C2: 10M synthetic docs, 3-term OR BM25 top-10, one shard per GPU", "docs_per_shard": args.docs,
                        "queries_per_step": args.queries, "k": 10, "vocabulary": 4096, "result_type": "Topk", "strategy": "auto (pruned top-k; exhaustive scan reported beside it)",
                        "value_counts": "queries x shards scanned (one 10M-doc shard per GPU)"} if is_bm else
                       {"workload": "C3: 10M x 768 f32, batch-64 cosine top-100 brute force", "rows_per_shard": args.rows,
                        "dim": args.dim, "batch": 64, "k": 100}),
            "global_qps": prim["qps"],
            "roofline": prim["roofline"],
            "cpu_baseline": prim.get("cpu_baseline"),
            "latency_ms": prim["latency_ms"],
        }
        if is_bm:
            line["exhaustive"] = bm["exhaustive"]
            line["topk_count"] = bm["topk_count"]
            line["bm25"] = {"build_s": bm["build_s"], "postings": int(bm["info"]["n_postings"]), "avgdl": bm["info"]["avgdl"],
                            "mean_algorithmic_bytes_per_query": bm["mean_bytes_per_query"], "mean_union_size": bm["mean_union"]}
        if vec is not None and is_bm:
            line["vector"] = {"metric": "queries/sec (cosine top-100, 10M x 768 f32, batch 64)", "value": vec["qps"] * world,
                              "global_qps": vec["qps"], "ms_per_step": vec["ms_per_step"], "roofline": vec["roofline"],
                              "cpu_baseline": vec.get("cpu_baseline"), "latency_ms": vec["latency_ms"], "build_s": vec["build_s"],
                              "rows_per_shard": args.rows, "dim": args.dim, "hybrid": vec.get("hybrid"), "ann": vec.get("ann"),
                              "i8": vec.get("i8")}
        elif vec is not None:
            line["i8"] = vec.get("i8")
        print(json.dumps(line), flush=True)
    sh.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
