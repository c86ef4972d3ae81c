# partitions per query (SS_BM25_P) of the exhaustive 16-bit scan on C2: kernel time per 1000-query call
for P in ${1:-4 5 6 7 8 10}; do SS_BM25_P=$P python tools/probes/exh_time.py P$P 2>&1 | grep -v amdgpu; done
