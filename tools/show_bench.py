"""prints the figures of a bench.py JSON line that one looks at first:  python tools/show_bench.py gpurun_out/x_bench.json"""
import json, sys
t = open(sys.argv[1]).read()
d = json.loads([l for l in t.splitlines() if l.startswith("{")][-1])
def g(o, *ks):
    for k in ks:
        if o is None: return None
        o = o.get(k)
    return o
f = lambda x: "None" if x is None else ("%.4g" % x if isinstance(x, float) else str(x))
print("value", f(d["value"]), "q/s  ms/step", f(d["ms_per_step"]), " end_to_end", f(d.get("value_end_to_end")))
r = d["roofline"]; print("  roofline: achieved", f(r["achieved"]), r["unit"], "frac", f(r["frac"]), "avg_launch_ms", f(r["avg_launch_ms"]), "alg GB/s", f(r.get("effective_GBs_on_algorithmic_bytes")))
e = d.get("exhaustive"); 
if e: print("exhaustive", f(e["value"]), "q/s  roofline", f(e["roofline"]["achieved"]), "GB/s frac", f(e["roofline"]["frac"]), "kernel ms", f(e["roofline"]["avg_launch_ms"]))
tc = d.get("topk_count") or {}
print("topk_count", {k: f(v["value"]) for k, v in tc.items() if isinstance(v, dict)})
print("intersection", f(g(d, "intersection", "value")))
print("latency", {k: f(v) for k, v in d["latency_ms"].items() if k != "clock"})
print("e2e", {k: f(v) for k, v in (d.get("end_to_end") or {}).items() if k not in ("entry_point", "unit")})
print("rationed", f(g(d, "rationed_vocabulary", "value")), "churn", f(g(d, "rationed_vocabulary", "churn_value")), " multi_field", {k: f(v["value"]) for k, v in (d.get("multi_field") or {}).items() if isinstance(v, dict)})
rv = d.get("realistic_vocabulary")
if rv: print("vocabulary", rv["vocabulary"], "terms:", f(rv["value"]), "q/s; dense parts only", f(rv["same_queries_without_their_rare_terms"]), "; rare-term queries", f(rv["queries_naming_a_rare_term"]),
             "sparse MB", rv["sparse_tier_bytes"] >> 20, "vs directory if dense MB", rv["directory_bytes_if_dense"] >> 20, "gen/append s", f(rv["host_generation_s"]), f(rv["append_s"]))
print("cpu_baseline", f(g(d, "cpu_baseline", "value")), "cores", g(d, "cpu_baseline", "cores"))
v = d.get("vector")
if v:
    print("vector", f(v["value"]), "q/s ms/call", f(v["ms_per_call"]), "roofline", f(v["roofline"]["achieved"]), v["roofline"]["unit"], f(v["roofline"]["frac"]))
    print("  latency", {k: f(x) for k, x in v["latency_ms"].items() if k != "clock"})
    print("  hybrid", f(g(v, "hybrid", "value")), " ann", {k: f(x) for k, x in (v.get("ann") or {}).items() if "ms" in k})
    i8 = v.get("i8") or {}
    print("  i8", f(i8.get("value")), "frac", f(g(i8, "roofline", "frac")), "ann", {k: f(x) for k, x in (i8.get("ann") or {}).items() if "ms" in k})
    print("  cpu", f(g(v, "cpu_baseline", "value")), g(v, "cpu_baseline", "rows"))
for k, x in (d.get("sharded") or {}).items():
    if isinstance(x, dict): print("sharded", k, f(x["value"]), "q/s ms/call", f(x["ms_per_call"]), "allgather_us", f(x["allgather_us"]))
for k, x in (d.get("concurrent_callers") or {}).items():
    if isinstance(x, dict): print("concurrent", k, f(x["value"]), "q/s p50", f(x["latency_us_p50"]), "p99", f(x["latency_us_p99"]), "batch", f(x["mean_lexical_batch"]), f(x["mean_vector_batch"]))
print("parity", {k: (v["queries"], round(v.get("seconds", 0.0), 1)) for k, v in (d.get("parity_full_size") or {}).items()})
