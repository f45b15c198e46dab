#!/bin/bash
# round 5, GPU call Z: do the walk kernels of the pipelined product-quantised search (COLTT_PQ_STREAMS=3) overlap?  kernel trace, start / end per launch
mkdir -p gpurun_out/r05z
O=$PWD/gpurun_out/r05z
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
COLTT_PQ_STREAMS=3 PROBE_PLAIN=0 timeout 300 rocprofv3 --kernel-trace -f csv -d /tmp/kt -o kt -- python $R/tools/hnswpq_probe.py 10000000 64:32 1408 0 > $O/trace.out 2> $O/trace.err
python3 - > $O/timeline.txt <<'PY'
import csv, glob
f = glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "hnsw_pq" in r["Kernel_Name"] or "pq_lut" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-45:]   # the timed call: 10 groups x (table, walk, re-rank, select)
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    n = r["Kernel_Name"]; n = n[n.find("hnsw_pq"):][:28] if "hnsw_pq" in n else "pq_lut_kernel"
    print(f"{n:30s} queue {str(r.get('Queue_Id','?')):>3s} grid {str(r.get('Grid_Size_X', r.get('Grid_Size','?'))):>8s} start {(int(r['Start_Timestamp'])-t0)/1e6:8.3f} ms  end {(int(r['End_Timestamp'])-t0)/1e6:8.3f} ms")
PY
cat $O/timeline.txt | head -60
