"""CPU-side tests (-m "not gpu"): the C-ABI library loads and exports every symbol include/seekstorm_hip.h declares,
the host-side planner logic (merge / RRF / idf / normalisation / error behaviour) agrees with the oracle, and the
multi-process gather + merge path is exercised with gloo, world_size 2.  No GPU compute is called here."""
import os
import re
import socket

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import ctypes as C
    from seekstorm_amd import _native as N
    hdr = open(os.path.join(ROOT, "include", "seekstorm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(ss_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    L = C.CDLL(N.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), f"{name} declared in seekstorm_hip.h but not exported"
    bound = {s[0] for s in N.SYMBOLS}
    assert set(declared) == bound, (set(declared) ^ bound)
    assert N.lib().ss_abi_version() == 7
    assert N.lib().ss_strerror(-4).decode().startswith("not answered by the device path")


def test_driver_build_entry_point():
    """__graft_entry__.build() is what the driver runs as its "does it build" check: it must pass on the tree as it is
    (round 6: it still asserted the previous ABI version)."""
    import importlib, sys
    sys.path.insert(0, ROOT)
    G = importlib.import_module("__graft_entry__")
    G.build()


def test_struct_layout_matches_header():
    import ctypes as C
    from seekstorm_amd import _native as N
    assert C.sizeof(N.Bm25Query) == 4 + 4 + 4 * 32 + 4 * 32 + 4 + 12  # SS_MAX_QUERY_TERMS = 32 (ABI v5)
    assert N.BM25_QUERY_DTYPE.fields["term"][1] == 8 and N.BM25_QUERY_DTYPE.fields["idf"][1] == 8 + 4 * 32
    # the structs the header declares, as gcc lays them out
    import os, subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "seekstorm_hip.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu %zu\\n", '
           'sizeof(ss_bm25_query), sizeof(ss_ann_mode), offsetof(ss_ann_mode, field_mask), sizeof(ss_facet_filter), '
           'offsetof(ss_facet_filter, lo), offsetof(ss_facet_filter, values), sizeof(ss_ref_block)); return 0; }\n')
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-std=c11", "-I", os.path.join(root, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")], check=True)
        got = [int(x) for x in subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.split()]
    assert got == [C.sizeof(N.Bm25Query), C.sizeof(N.AnnModeC), N.AnnModeC.field_mask.offset, C.sizeof(N.FacetFilterC),
                   N.FacetFilterC.lo.offset, N.FacetFilterC.values.offset, C.sizeof(N.RefBlock)]


def test_rust_ffi_file_is_the_header():
    """integration/hip_ffi.rs (the extern "C" side of the Rust binding) is generated from the header: the
    committed file equals a fresh generation, names every entry point, and its #[repr(C)] structs have the sizes and field
    offsets gcc gives the C structs (computed with the C layout rules from the Rust field types)."""
    import re, subprocess, sys, tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_ffi as G
    text, names = G.generate()
    assert open(os.path.join(ROOT, "integration", "hip_ffi.rs")).read() == text, "run python tools/gen_rust_ffi.py"
    hdr = open(os.path.join(ROOT, "include", "seekstorm_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(ss_[a-z0-9_]+)\s*\(", hdr)))
    assert sorted(names) == declared
    assert text.count("{") == text.count("}") and text.count("(") == text.count(")")
    size = {"u8": 1, "i8": 1, "u16": 2, "i16": 2, "u32": 4, "i32": 4, "f32": 4, "c_int": 4, "u64": 8, "i64": 8, "f64": 8}
    consts = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"pub const (SS_\w+): \w+ = (-?\w+);", text)}
    layouts = {}
    for m in re.finditer(r"pub struct (Ss\w+) \{\n((?:    pub [^\n]+\n)+)\}", text):
        off, align, fields = 0, 1, []
        for f in re.finditer(r"pub (\S+): ([^,]+),", m.group(2)):
            ty, n = f.group(2).strip(), 1
            a = re.fullmatch(r"\[(\w+); (\w+)(?: as usize)?\]", ty)
            if a:
                ty, n = a.group(1), consts[a.group(2)] if a.group(2) in consts else int(a.group(2))
            sz = 8 if ty.startswith("*") else size[ty]
            off = (off + sz - 1) // sz * sz
            fields.append((f.group(1).replace("r#", ""), off))
            off += sz * n
            align = max(align, sz)
        layouts[m.group(1)] = ((off + align - 1) // align * align, fields)
    cname = {"SsRefBlock": "ss_ref_block", "SsBm25Query": "ss_bm25_query", "SsFacetPoint": "ss_facet_point", "SsFacetFilter": "ss_facet_filter",
             "SsAnnMode": "ss_ann_mode", "SsVecLevel": "ss_vec_level", "SsResultSort": "ss_result_sort"}
    assert set(layouts) == set(cname)
    items, want = [], []
    for rs, (total, fields) in layouts.items():
        items.append("sizeof(%s)" % cname[rs]); want.append(total)
        for fname, off in fields:
            items.append("offsetof(%s, %s)" % (cname[rs], fname)); want.append(off)
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "seekstorm_hip.h"\nint main(void) { size_t v[] = {%s}; '
           'for (unsigned i = 0; i < sizeof v / sizeof v[0]; i++) printf("%%zu ", v[i]); return 0; }\n' % ", ".join(items))
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")], check=True)
        got = [int(x) for x in subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.split()]
    assert got == want


def test_no_gpu_means_loud_failure_not_fallback():
    import ctypes as C
    import seekstorm_amd as S
    n = C.c_int(-1)
    rc = S.lib().ss_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(S.SeekStormHipError):
        S.Shard(0)


def test_commit_entry_points_refuse_missing_arguments():
    """the commit seam's entry points (ABI v4) check their arguments before they touch a device: no shard / no offsets -> SS_EINVAL (-1)"""
    import ctypes as C
    from seekstorm_amd import _native as N
    L = N.lib()
    offs = np.zeros(2, np.uint64)
    assert L.ss_bm25_append_sparse_level(None, 0, 1, N.ptr(offs, N.u64p), None, None, None, None, 0) == -1
    assert L.ss_bm25_append_level(None, 0, 1, None, 1, N.ptr(offs, N.u64p), None, None) == -1
    assert L.ss_index_bin_decode_all(None, None, None, None, 0, None, None, 0) == -1
    assert L.ss_index_bin_decode_stats(None, 0, None, None) == -1
    assert L.ss_vec_append_rows(None, None) == -1


def test_commit_level_splits_a_level_between_the_tiers():
    """Shard.commit_level (the mirror of the commit seam): one CSR of all known terms in id order -> the dense terms' part for
    ss_bm25_append_level, the rare terms' part (offsets rebased, positions pool cut where the dense postings' positions end) for
    ss_bm25_append_sparse_level.  Host logic only: the two ABI calls are captured."""
    import seekstorm_amd as S
    calls = {}

    class Capture(S.Shard):
        def __init__(self):  # no device
            pass

        def append_level(self, level, level_doclen, offs, docs, tfs, positions=None, npos=None):
            calls["dense"] = (level, np.asarray(offs), np.asarray(docs), np.asarray(tfs), positions, npos)

        def append_sparse_level(self, level, offs, docs, tfs, positions=None, npos=None):
            calls["sparse"] = (level, np.asarray(offs), np.asarray(docs), np.asarray(tfs), positions, npos)

    offs = np.array([0, 2, 3, 3, 5, 6], np.uint64)            # 5 terms: 3 dense, 2 rare
    docs = np.array([70000, 70010, 70001, 70002, 70020, 70005], np.uint32)
    tfs = np.array([2, 1, 3, 1, 2, 1], np.uint16)
    pos = np.arange(int(tfs.sum()), dtype=np.uint16)
    sh = Capture()
    sh.commit_level(1, np.zeros(100, np.uint8), offs, docs, tfs, n_dense_terms=3, positions=pos)
    lv, o, d, t, p, n = calls["dense"]
    assert lv == 1 and o.tolist() == [0, 2, 3, 3] and d.tolist() == [70000, 70010, 70001] and t.tolist() == [2, 1, 3] and p.tolist() == list(range(6))
    lv, o, d, t, p, n = calls["sparse"]
    assert lv == 1 and o.tolist() == [0, 2, 3] and d.tolist() == [70002, 70020, 70005] and t.tolist() == [1, 2, 1] and p.tolist() == [6, 7, 8, 9]
    # counts from npos where given (n-gram components): the cut follows them
    npos = np.array([2, 0, 3, 1, 0, 1], np.uint16)
    sh.commit_level(1, np.zeros(100, np.uint8), offs, docs, tfs, n_dense_terms=3, positions=np.arange(7, dtype=np.uint16), npos=npos)
    assert calls["dense"][4].tolist() == [0, 1, 2, 3, 4] and calls["dense"][5].tolist() == [2, 0, 3]
    assert calls["sparse"][4].tolist() == [5, 6] and calls["sparse"][5].tolist() == [1, 0, 1]
    calls.clear()
    sh.commit_level(0, np.zeros(10, np.uint8), offs[:4], docs[:3], tfs[:3])  # no tier: every term is dense
    assert "sparse" not in calls and calls["dense"][1].tolist() == [0, 2, 3, 3]


def test_product_sources_do_not_reference_oracle_code():
    pkg = os.path.join(ROOT, "seekstorm_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "ss_oracle.h" not in txt and "import oracle" not in txt and "from oracle" not in txt, f


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_merge_results_matches_oracle(mode):
    import seekstorm_amd as S
    from oracle import oracle as O
    rng = np.random.default_rng(7 + mode)
    for trial in range(20):
        nl, nv = int(rng.integers(0, 40)), int(rng.integers(0, 40))
        ld = rng.choice(200, nl, replace=False).astype(np.uint64)
        vd = rng.choice(200, nv, replace=False).astype(np.uint64)
        ls = np.sort(rng.random(nl).astype(np.float32) * 10)[::-1].copy()
        vs = np.sort(rng.random(nv).astype(np.float32))[::-1].copy()
        off, length = int(rng.integers(0, 5)), int(rng.integers(1, 30))
        d, s, src = S.merge_results(mode, (ld, ls), (vd, vs), off, length)
        od, os_, osrc = O.merge(mode, (ld, ls), (vd, vs), off, length)
        assert np.array_equal(d, od) and np.allclose(s, os_, rtol=1e-6) and np.array_equal(src, osrc)


def test_rrf_semantics():
    import seekstorm_amd as S
    # search.rs:1962-2035: k = 0.6, ranks from 0; doc in both lists sums and becomes Hybrid
    d, s, src = S.merge_results(S.SearchMode.Hybrid, ([10, 11, 12], [3.0, 2.0, 1.0]), ([11, 20], [0.9, 0.8]), 0, 10)
    assert list(d) == [11, 10, 20, 12]
    assert np.allclose(s, [1 / 1.6 + 1 / 0.6, 1 / 0.6, 1 / 1.6, 1 / 2.6], rtol=1e-6)
    assert list(src) == [S.ResultSource.Hybrid, S.ResultSource.Lexical, S.ResultSource.Vector, S.ResultSource.Lexical]
    # offset / length applied after the merge (search.rs:2109-2119)
    d2, _, _ = S.merge_results(S.SearchMode.Hybrid, ([10, 11, 12], [3.0, 2.0, 1.0]), ([11, 20], [0.9, 0.8]), 1, 2)
    assert list(d2) == [10, 20]


def test_host_scalars_match_oracle():
    import seekstorm_amd as S
    from oracle import oracle as O
    L = O.lib()
    for N_, n in [(1_000_000, 1000), (10_000_000, 125_000), (300_000, 60_000), (4, 2)]:
        assert abs(float(S.idf_f32(N_, n)) - L.so_idf(N_, n)) <= 2e-7 * max(1.0, L.so_idf(N_, n))
    v = O.vec_gen(3, 0, 1, 768, normalize=False)[0]
    assert np.allclose(S.normalize_f32(v), O.normalize(v), rtol=0, atol=1e-7)
    assert abs(S.threshold_raw(0.7) - L.so_threshold_raw(0.7)) < 1e-2
    assert S.threshold_raw(None) < -3e38


# ------------------------------------------------------------------ world_size-2 gloo: gather + merge == single-process merge
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nq, k, seed, q):
    import torch
    import torch.distributed as dist
    from seekstorm_amd import distributed as D
    from seekstorm_amd.search import SearchMode
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lists = _fake_shard_lists(world, nq, k, seed)
    lex, vec = lists[rank]
    out = {}
    for name, (doc, score, cnt), mode in (("lex", lex, SearchMode.Lexical), ("vec", vec, SearchMode.Vector)):
        g = D.all_gather_topk(torch.from_numpy(doc), torch.from_numpy(score), torch.from_numpy(cnt))
        # the single packed collective the GPU path uses carries the same three arrays
        gp = D.unpack_gathered(D.all_gather_topk_packed(torch.from_numpy(doc), torch.from_numpy(score), torch.from_numpy(cnt)), nq, k)
        assert all(torch.equal(a, b) for a, b in zip(g, gp))
        out[name] = D.merge_gathered_host(*g, 0, k, mode)
        out[name + "_g"] = g
    out["hyb"] = D.merge_gathered_hybrid_host(out.pop("lex_g"), out.pop("vec_g"), 0, k)
    q.put((rank, {kk: [(np.asarray(a).tolist(), np.asarray(b).tolist()) for a, b, *_ in v] for kk, v in out.items()}))
    dist.barrier()
    dist.destroy_process_group()


def _fake_shard_lists(world, nq, k, seed):
    """deterministic per-shard top-k lists (sorted desc, ragged counts) for both modes"""
    rng = np.random.default_rng(seed)
    res = []
    for s in range(world):
        per = []
        for scale in (10.0, 1.0):
            doc = np.stack([rng.choice(5000, k, replace=False) for _ in range(nq)]).astype(np.int32)
            score = np.sort(rng.random((nq, k)).astype(np.float32) * scale, axis=1)[:, ::-1].copy()
            cnt = rng.integers(k // 2, k + 1, nq).astype(np.int32)
            per.append((doc, score, cnt))
        res.append(tuple(per))
    return res


def test_two_process_gloo_gather_merge_equals_single_process():
    import torch.multiprocessing as mp
    from oracle import oracle as O
    world, nq, k, seed = 2, 5, 16, 99
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nq, k, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    lists = _fake_shard_lists(world, nq, k, seed)
    assert got[0] == got[1]  # identical merge on every rank
    for qi in range(nq):
        cat = {}
        for mi, name in enumerate(("lex", "vec")):
            ids, sc = [], []
            for s in range(world):
                doc, score, cnt = lists[s][mi]
                n = int(cnt[qi])
                ids += [int(x) * world + s for x in doc[qi, :n]]
                sc += [float(x) for x in score[qi, :n]]
            cat[name] = (ids, sc)
            od, os_, _ = O.merge(mi, cat[name], cat[name], 0, k) if mi == 0 else O.merge(1, None, cat[name], 0, k)
            gd, gs = got[0][name][qi]
            assert gd == [int(x) for x in od] and np.allclose(gs, os_, rtol=1e-6)
        od, os_, _ = O.merge(2, cat["lex"], cat["vec"], 0, k)
        gd, gs = got[0]["hyb"][qi]
        assert gd == [int(x) for x in od] and np.allclose(gs, os_, rtol=1e-6)


# ------------------------------------------------------------------ world_size-2 gloo: the ss_*_search_sharded exchange protocol
def _worker_exchange(rank, world, port, nq, k, seed, fail_rank, q):
    import torch
    import torch.distributed as dist
    from seekstorm_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lists = _fake_shard_lists(world, nq, k, seed)
    (ld, ls, lc), (vd, vs, vc) = lists[rank]
    rng = np.random.default_rng(seed + 100 * rank)
    lt, vt = rng.integers(0, 10**6, nq), rng.integers(0, 10**6, nq)
    T = torch.from_numpy
    res = {}
    hyb, tot = D.search_hybrid_exchanged((T(ld), T(ls), T(lc)), (T(vd), T(vs), T(vc)), lt, vt, 2, k - 2)
    res["hyb"] = [(np.asarray(a).tolist(), np.asarray(b).tolist()) for a, b, *_ in hyb]
    res["tot"] = tot.tolist()
    res["own"] = (lt.tolist(), vt.tolist())
    # a rank whose own search failed still enters the collective; EVERY rank then reports the failure, none blocks
    try:
        D.search_hybrid_exchanged((T(ld), T(ls), T(lc)), (T(vd), T(vs), T(vc)), lt, vt, 0, k, local_error=-5 if rank == fail_rank else 0)
        res["fail"] = "no error"
    except D.PeerError as e:
        res["fail"] = str(e)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_gloo_hybrid_exchange_in_one_gather():
    """Index-level hybrid merge over two ranks through the one-gather protocol of ss_hybrid_search_sharded (lists of both modes,
    totals and a status word in ONE collective): equal to the oracle's merge of the concatenated lists on every rank; totals =
    sum over the shards of max(lexical, vector); a failing rank makes every rank fail instead of hanging the others"""
    import torch.multiprocessing as mp
    from oracle import oracle as O
    world, nq, k, seed = 2, 4, 12, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_exchange, args=(r, world, port, nq, k, seed, 1, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0]["hyb"] == got[1]["hyb"] and got[0]["tot"] == got[1]["tot"]
    lists = _fake_shard_lists(world, nq, k, seed)
    for qi in range(nq):
        cat = []
        for mi in range(2):
            ids, sc = [], []
            for s in range(world):
                doc, score, cnt = lists[s][mi]
                n = int(cnt[qi])
                ids += [int(x) * world + s for x in doc[qi, :n]]
                sc += [float(x) for x in score[qi, :n]]
            cat.append((ids, sc))
        od, os_, _ = O.merge(2, cat[0], cat[1], 2, k - 2)
        gd, gs = got[0]["hyb"][qi]
        assert gd == [int(x) for x in od] and np.allclose(gs, os_, rtol=1e-6)
        assert got[0]["tot"][qi] == sum(max(got[r]["own"][0][qi], got[r]["own"][1][qi]) for r in range(world))
    assert "local search failed" in got[1]["fail"] and "peer" in got[0]["fail"]


def test_ann_mode_struct_matches_the_header():
    """ss_ann_mode grew flags / reserved in ABI v3: the ctypes mirror, the header and the generated Rust declaration agree"""
    import ctypes as C
    import re
    from seekstorm_amd import _native as N
    assert C.sizeof(N.AnnModeC) == 24 and N.AnnModeC.flags.offset == 16 and N.AnnModeC.field_mask.offset == 8
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "seekstorm_hip.h")).read()
    body = hdr[hdr.index("typedef struct ss_ann_mode {"):hdr.index("} ss_ann_mode;")]
    names = re.findall(r"^\s*(?:uint32_t|uint64_t|float)\s+(\w+);", body, re.M)
    assert names == ["n_probe", "cluster_threshold_raw", "field_mask", "flags", "reserved"]
    assert int(re.search(r"#define SS_ANN_REPORT_OBSERVED (\d+)u", hdr).group(1)) == N.SS_ANN_REPORT_OBSERVED
    rs = open(os.path.join(root, "integration", "hip_ffi.rs")).read()
    assert "flags" in rs[rs.index("struct SsAnnMode"):rs.index("struct SsAnnMode") + 400]


def test_bench_compact_line_keeps_the_contract_and_its_bound():
    """bench.py's last stdout line: built from the FULL line of the round's final run (profiles/r5_bench_details.json),
    and from the same line blown up with junk legs -- one parseable object under 4096 bytes either way, the contract's keys whole."""
    import json
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r5_bench_details.json")))
    for extra in (0, 400):
        line = dict(full)
        if extra:  # more callers' legs than any run holds: sections are dropped before the bound is exceeded, the contract never
            line["concurrent_callers"] = dict(line.get("concurrent_callers") or {})
            for i in range(extra):
                line["concurrent_callers"][f"junk_{i}"] = {"value": 1.0 * i, "threads": i, "latency_us_p50": 1.0, "latency_us_p99": 2.0}
        out = bench.compact_line(line)
        txt = json.dumps(out)
        assert len(txt.encode()) <= 4096, len(txt)
        back = json.loads(txt)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                  "roofline", "cpu_baseline"):
            assert k in back, k
        assert back["vs_baseline"] is None and back["data"] == "synthetic" and "workload" in back["config"] and "model" not in back["config"]
        rf = back["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in rf, k
        assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in back["cpu_baseline"], k
        assert back["cpu_baseline"]["kind"] in ("port", "reference")


def test_committed_bench_line_and_counter_file_belong_together():
    """the line of the round's final run (profiles/<pmc_traffic.json's collected_as>_bench_line.json); its `traffic` figures come from
    profiles/pmc_traffic.json, which bench.py only uses while it was collected on the kernel sources of the tree (kernel_source_hash)."""
    import json
    import bench
    tag = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["collected_as"]
    line = json.load(open(os.path.join(ROOT, "profiles", tag + "_bench_line.json")))
    assert line["n_gpus"] == 1 and line["unit"] == "queries/s" and line["value"] > 1e6
    assert len(json.dumps(line).encode()) <= 4096
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for leg in ("bm25", "bm25_pruned", "vector", "vector_i8"):
        assert pmc[leg]["hbm_bytes_per_launch"] > 0 and pmc[leg]["launches"] > 0
    assert abs(line["roofline"]["traffic"] - pmc["bm25"]["hbm_bytes_per_launch"]) / pmc["bm25"]["hbm_bytes_per_launch"] < 1e-3
    if pmc["kernel_source_hash"] != bench.kernel_source_hash():
        pytest.skip("profiles/pmc_traffic.json predates the kernel sources of this tree: bench.py reports traffic = null until tools/collect_pmc.sh is re-run")


def test_bench_offers_the_one_process_shape():
    """bench.py --one-process S: the reference's own shape (one process, one task per shard; VERDICT r5 "next" 9) beside the rank-per-GPU form"""
    import subprocess, sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--one-process" in out.stdout and "--gpus" in out.stdout


def test_bench_stdout_carries_one_line():
    """bench.py's contract is ONE JSON line on stdout: whatever a library writes to file descriptor 1 after claim_stdout() (RCCL's version
    banner, through C stdio) lands on stderr, and print_line() puts the result on the real stdout."""
    import subprocess, sys
    code = ("import ctypes, os, sys; sys.path.insert(0, %r); import bench; bench.claim_stdout(); "
            "ctypes.CDLL(None).puts(b'banner through C stdio'); os.write(1, b'raw write to fd 1\\n'); print('python print'); "
            "bench.print_line('{\"x\": 1}')") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout == '{"x": 1}\n', repr(out.stdout)
    assert "banner through C stdio" in out.stderr and "raw write to fd 1" in out.stderr and "python print" in out.stderr
