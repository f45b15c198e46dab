#!/bin/bash
# round 5, GPU call U: where the product-quantised search's time goes (kernel trace: table / walk / re-rank / select) and what a partial re-rank costs in recall
mkdir -p gpurun_out/r05u
O=$PWD/gpurun_out/r05u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PROBE_PLAIN=0 PROBE_OUT=$O/trace_probe.jsonl timeout 500 rocprofv3 --kernel-trace --stats -f csv -d /tmp/kt -o kt -- python $R/tools/hnswpq_probe.py 10000000 64:32 1408 0 > $O/trace.out 2> $O/trace.err
find /tmp/kt -name "*kernel_stats*" -exec cp {} $O/kernel_stats.csv \;
find /tmp/kt -name "*kernel_trace*" -exec python3 - {} $O/walk_launches.txt \; <<'PY' 2>/dev/null
PY
python3 - <<'PY' > $O/pq_kernels.txt 2>&1
import csv, glob, collections
f = glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "pq" in n or "lut" in n.lower():
        agg[n[:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for n, v in agg.items():
    print(f"{n:92s} launches {len(v):3d}  last {v[-1]:9.3f} ms  max {max(v):9.3f} ms")
PY
cat $O/pq_kernels.txt | cut -c1-200
cd $R
PROBE_PLAIN=0 PROBE_OUT=$O/rerank.jsonl timeout 400 python tools/hnswpq_probe.py 10000000 64:32 1280,1408 0,768,512,384,256 > $O/rerank.out 2> $O/rerank.err
python - $O/rerank.jsonl <<'PY'
import sys, json
for l in open(sys.argv[1]):
    r = json.loads(l)
    if r.get("kind") == "pq": print("pq ef", r.get("ef"), "rerank", r.get("rerank"), "recall", r.get("recall"), "qps", r.get("qps"), "ms", r.get("kernel_ms"), r.get("error", ""))
PY
