"""Times the reference-structured CPU baseline (oracle so_search_lex_ref / so_bench_lex) on a sample of the C2 queries.
python tools/probes/cpu_baseline_c2.py [n_docs] [n_sample] [shards] [seconds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import oracle as O
import bench
from concurrent.futures import ThreadPoolExecutor

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 16
S = int(sys.argv[3]) if len(sys.argv) > 3 else (os.cpu_count() or 1)
secs = float(sys.argv[4]) if len(sys.argv) > 4 else 4.0
cores = os.cpu_count() or 1
tl, th = bench.make_c2_queries(O, 1000)
sample = tl[:ns]
voc = sorted({t for q in sample for t in q})
t0 = time.time()
dl = O.lex_doclen(n_docs)
with ThreadPoolExecutor(cores) as ex:
    parts = list(ex.map(lambda t: O.lex_term(t, th[t], n_docs), voc))
offs = np.zeros(len(voc) + 1, np.uint64); offs[1:] = np.cumsum([len(p[0]) for p in parts])
docs = np.concatenate([p[0] for p in parts]); tfs = np.concatenate([p[1] for p in parts])
print("gen %.1fs, %d postings" % (time.time() - t0, len(docs)))
remap = {t: i for i, t in enumerate(voc)}
qs = np.array([[remap[t] for t in q] for q in sample], np.uint32)
one = O.Shard(n_docs, dl, offs, docs, tfs)
for q in qs[:3]:
    t0 = time.time(); a = one.search_ref(q, O.OP_OR, 10, O.RT_TOPK); t1 = time.time(); b = one.search(q, O.OP_OR, 10, O.RT_TOPK); t2 = time.time()
    print("union_docid_3 %.2f ms   union_scan %.2f ms   equal=%s" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, np.allclose(a[1], b[1])))
qps, done, lat = O.bench_lex([one], qs, O.OP_OR, 10, O.RT_TOPK, 1, 1, secs)
print("1 shard, 1 thread: %.1f q/s, p50 %.0f us p99 %.0f us" % (qps, *np.percentile(lat, [50, 99])))
qps, done, lat = O.bench_lex([one], qs, O.OP_OR, 10, O.RT_TOPK, 0, cores, secs)
print("1 shard, %d threads throughput: %.1f q/s" % (cores, qps))
t0 = time.time()
shards = [O.Shard(*x) for x in O.split_corpus(n_docs, dl, offs, docs, tfs, S)]
print("split + build %d shards %.1fs" % (S, time.time() - t0))
qps, done, lat = O.bench_lex(shards, qs, O.OP_OR, 10, O.RT_TOPK, 0, cores, secs)
print("S=%d throughput mode (%d threads): %.1f q/s" % (S, cores, qps))
qps, done, lat = O.bench_lex(shards, qs, O.OP_OR, 10, O.RT_TOPK, 1, S, secs)
print("S=%d latency mode: %.1f q/s, p50 %.0f us p99 %.0f us" % (S, qps, *np.percentile(lat, [50, 99])))
