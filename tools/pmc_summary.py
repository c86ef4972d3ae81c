#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc databases collected by `tools/collect_pmc.sh` into profiles/<tag>_pmc_summary.md and
profiles/pmc_traffic.json (HBM bytes per launch of the dominant kernels, corrected as MI355X_MICROARCH.md section
HBM prescribes: FETCH_SIZE/WRITE_SIZE are in KiB and FETCH_SIZE reports exactly half of a wide coalesced read on
gfx950 -> bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024).
Usage: python tools/pmc_summary.py <gpurun_out dir> <tag> <full_bm25_launches> <vec_passes>"""
import json
import os
import sqlite3
import sys
from collections import defaultdict


def totals(db, pat):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    ix = {c: i for i, c in enumerate(cols)}
    kn = "kernel_name" if "kernel_name" in ix else "name"
    agg = defaultdict(float)
    disp = set()
    for r in cur.execute("select * from counters_collection"):
        if pat in r[ix[kn]]:
            agg[r[ix["counter_name"]]] += float(r[ix["value"]])
            disp.add(r[ix["dispatch_id"]])
    return dict(agg), len(disp)


def main():
    d, tag, n_bm, n_vec = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    out = {}
    lines = [f"# PMC summary ({tag})", "",
             "rocprofv3 --pmc <counters> --kernel-trace -- python bench.py --no-cpu --steps 4 --warmup 1 (separate passes per counter group).",
             f"Per-launch figures divide the kernel totals by the number of full-size launches in that run "
             f"({n_bm} BM25 batches of 1000 queries; {n_vec} vector passes of 64 queries = 7 row-chunk launches each); "
             "the 60 single-query latency probes add < 0.4 % to the BM25 totals.", ""]
    for kern, pat, n in (("bm25", "bm25_scan", n_bm), ("vector", "vec_scan_kernel", n_vec)):
        f, _ = totals(os.path.join(d, "pmc_FETCH_SIZE", "x_results.db"), pat)
        w, _ = totals(os.path.join(d, "pmc_WRITE_SIZE", "x_results.db"), pat)
        s, nd = totals(os.path.join(d, "pmc_SQ_VALU_MFMA_BUSY_CYCLES", "x_results.db"), pat)
        fetch = f.get("FETCH_SIZE", 0.0)
        write = w.get("WRITE_SIZE", 0.0)
        hbm = (2.0 * fetch + write) * 1024.0 / n
        out[kern] = {"hbm_bytes_per_launch": hbm, "fetch_size_kib_total": fetch, "write_size_kib_total": write, "launches": n}
        lines += [f"## {kern}: `{pat}`", "",
                  f"- FETCH_SIZE total {fetch:.4g} KiB, WRITE_SIZE total {write:.4g} KiB over {n} full launches",
                  f"- HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / {n} = **{hbm / 1e9:.3f} GB**"]
        gui = f.get("GRBM_GUI_ACTIVE", 0.0)
        if gui:
            lines.append(f"- GRBM_GUI_ACTIVE total {gui:.4g} (summed over the 8 XCDs) -> {gui / 8 / n:.4g} shader cycles per launch")
            out[kern]["cycles_per_launch"] = gui / 8 / n
        for c in sorted(s):
            lines.append(f"- {c}: total {s[c]:.4g}, per launch {s[c] / n:.4g}")
        if s.get("SQ_VALU_MFMA_BUSY_CYCLES") and gui:
            util = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * gui / 8)
            out[kern]["mfma_util"] = util
            lines.append(f"- **MFMA utilisation** = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x shader cycles) = **{100 * util:.1f} %** "
                         "(busy cycles = 64 per v_mfma_f32_32x32x2_f32)")
        if s.get("SQ_WAVE_CYCLES"):
            wc = s["SQ_WAVE_CYCLES"]
            lines.append(f"- wave-time split: active {100 * s.get('SQ_ACTIVE_INST_ANY', 0) / wc:.0f} %, s_waitcnt/barrier wait "
                         f"{100 * s.get('SQ_WAIT_ANY', 0) / wc:.0f} %, issue stall {100 * s.get('SQ_WAIT_INST_ANY', 0) / wc:.0f} %")
        lines.append("")
    os.makedirs("profiles", exist_ok=True)
    open(os.path.join("profiles", f"{tag}_pmc_summary.md"), "w").write("\n".join(lines) + "\n")
    json.dump(out, open(os.path.join("profiles", "pmc_traffic.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
