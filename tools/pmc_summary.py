#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc databases collected by `tools/collect_pmc.sh all <tag>` (separate passes per counter group:
fetch, write, sqA, sqB, sqC -- never combined with API tracing) into profiles/<tag>_pmc_summary.md and
profiles/pmc_traffic.json.

HBM bytes follow MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are KiB, and on gfx950 FETCH_SIZE reports
exactly half of the bytes read -> bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The factor was checked on kernels of known
traffic (tools/probes/pmc_calib.hip, profiles/r2_pmc_calibration.md): 2.00 for coalesced 16-byte streams AND for scattered
8-byte gathers (a gather moves a whole 128-byte line and the counter sees half of it); WRITE_SIZE needs no factor.
hbm_bytes_per_launch_low is the uncorrected reading (FETCH_SIZE + WRITE_SIZE), a floor.  BM25 figures are per FULL launch
(the dispatches with the largest grid: 1000-query batches; single-query latency probes are excluded); vector figures
are per 64-query pass (4 row-chunk launches of vec_scan_kernel / vec8_scan_kernel).
Usage: python tools/pmc_summary.py <gpurun_out dir> <tag>"""
import hashlib
import json
import os
import sqlite3
import sys
from collections import defaultdict


KERNEL_SOURCES = ("bm25.hip", "bm25_dev.h", "bm25_fast.hip", "bm25_probe.hip", "bm25_probe_body.h", "bm25_scan16.hip", "bm25_small.hip", "ss_common.h",
                  "vec8_scan.hip", "vec_scan.hip")


def kernel_source_hash():
    """the same hash bench.py computes: pmc_traffic.json is only believed for the kernel sources it was collected on"""
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "seekstorm_amd", "csrc")
    for f in KERNEL_SOURCES:  # the files that define the measured kernels (not the loaders, the ABI layer, the generators)
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def load(db, pat):
    """-> {dispatch_id: {'grid': g, 'dur': ns, counter: value}} for kernels whose name contains pat"""
    cur = sqlite3.connect(db).cursor()
    out = defaultdict(dict)
    for did, name, grid, dur, cn, val in cur.execute(
            "select dispatch_id, kernel_name, grid_size, duration, counter_name, value from counters_collection"):
        if pat in name:
            d = out[did]
            d["grid"], d["dur"] = grid, dur
            d[cn] = d.get(cn, 0.0) + float(val)
    return out


def main():
    d, tag = sys.argv[1], sys.argv[2]
    db = lambda g: os.path.join(d, f"pmc_{tag}_{g}", "x_results.db")
    out, lines = {}, [f"# PMC summary ({tag})", "",
                      "`rocprofv3 --pmc <group> --kernel-trace -- python bench.py --workload all --quick --no-topk-count --steps 4 --warmup 1`, "
                      "one run per counter group (tools/collect_pmc.sh).  SQ_* wave counters are in quad-cycles "
                      "(MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs.", ""]
    LAUNCHES_PER_PASS = 4  # chunk schedule of the vector scans: 16 tiles, then 16 x the prefix (10 M rows)
    for kern, pat in (("bm25", "bm25_scan16_kernel"), ("bm25_f32_scan", "bm25_scan_fast_kernel"), ("bm25_pruned", "bm25_probe_kernel"), ("bm25_union_count", "bm25_union_count_kernel"),
                      ("vector", "vec_scan_kernel<true, false>"), ("vector_small_batch", "vec_scan_kernel<false, false>"),
                      ("vector_i8", "vec8_scan_kernel<false, true, false>")):
        groups = {g: load(db(g), pat) for g in ("fetch", "write", "sqA", "sqB", "sqC") if os.path.exists(db(g))}
        if not groups.get("fetch"):
            continue

        def sel(g):
            rows = list(groups[g].values())
            if kern.startswith("bm25"):
                gmax = max(r["grid"] for r in rows)
                rows = [r for r in rows if r["grid"] == gmax]
                return rows, len(rows)
            return rows, max(1, len(rows) // LAUNCHES_PER_PASS)

        def tot(g, c):
            rows, n = sel(g)
            return sum(r.get(c, 0.0) for r in rows), n

        fetch, n = tot("fetch", "FETCH_SIZE")
        write, nw = tot("write", "WRITE_SIZE")
        gui, _ = tot("fetch", "GRBM_GUI_ACTIVE")
        hbm = (2.0 * fetch / n + write / nw) * 1024.0
        rows, _ = sel("fetch")
        dur_ms = sum(r["dur"] for r in rows) / n / 1e6
        unit = "full launch (1000 queries)" if kern.startswith("bm25") else f"64-query pass ({LAUNCHES_PER_PASS} launches)"
        out[kern] = {"hbm_bytes_per_launch": hbm, "hbm_bytes_per_launch_low": (fetch / n + write / nw) * 1024.0, "fetch_size_kib_total": fetch, "write_size_kib_total": write,
                     "launches": n, "cycles_per_launch": gui / 8 / n, "kernel_ms_per_launch_profiled": dur_ms}
        lines += [f"## {kern}: `{pat}` -- per {unit}, {n} of them", "",
                  f"- FETCH_SIZE {fetch / n:.4g} KiB, WRITE_SIZE {write / nw:.4g} KiB -> HBM traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 = "
                  f"**{hbm / 1e9:.3f} GB**",
                  f"- kernel time under the profiler {dur_ms:.3f} ms; GRBM_GUI_ACTIVE / 8 = {gui / 8 / n:.4g} shader cycles "
                  f"(effective clock {gui / 8 / n / dur_ms / 1e6:.2f} GHz)"]
        cu_cycles = gui / 8 / n * 256
        vals = {}
        for g in ("sqA", "sqB", "sqC"):
            if g not in groups:
                continue
            rows, ng = sel(g)
            names = sorted({c for r in rows for c in r if c not in ("grid", "dur")})
            for c in names:
                vals[c] = sum(r.get(c, 0.0) for r in rows) / ng
        for c in sorted(vals):
            if c != "GRBM_GUI_ACTIVE":
                lines.append(f"- {c}: {vals[c]:.4g}")
        if vals.get("SQ_WAVE_CYCLES"):
            wc = vals["SQ_WAVE_CYCLES"]
            lines.append(f"- wave-time split: issuing {100 * vals.get('SQ_ACTIVE_INST_ANY', 0) / wc:.0f} %, waiting on s_waitcnt / barrier "
                         f"{100 * vals.get('SQ_WAIT_ANY', 0) / wc:.0f} %, issue-stalled {100 * vals.get('SQ_WAIT_INST_ANY', 0) / wc:.0f} %")
        if vals.get("SQ_LDS_IDX_ACTIVE"):
            lines.append(f"- LDS busy {100 * vals['SQ_LDS_IDX_ACTIVE'] / cu_cycles:.0f} % of CU cycles, of which bank conflicts "
                         f"{100 * vals.get('SQ_LDS_BANK_CONFLICT', 0) / vals['SQ_LDS_IDX_ACTIVE']:.0f} %")
            out[kern]["lds_busy"] = vals["SQ_LDS_IDX_ACTIVE"] / cu_cycles
        if vals.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            util = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * gui / 8 / n)
            out[kern]["mfma_util"] = util
            lines.append(f"- **MFMA utilisation** = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x shader cycles) = **{100 * util:.1f} %** "
                         "(64 busy cycles per v_mfma_f32_32x32x2_f32)")
        lines.append("")
    out["kernel_source_hash"] = kernel_source_hash()
    out["collected_as"] = tag
    os.makedirs("profiles", exist_ok=True)
    open(os.path.join("profiles", f"{tag}_pmc_summary.md"), "w").write("\n".join(lines) + "\n")
    json.dump(out, open(os.path.join("profiles", "pmc_traffic.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
