#!/bin/bash
# round 5, GPU call AA: the walk's tables written as binary16 by their own kernel (4 KiB per query instead of 64 KiB of f32) — parity, then the kernel split again
mkdir -p gpurun_out/r05aa
O=$PWD/gpurun_out/r05aa
timeout 500 python -m pytest tests/test_gpu_round5.py tests/test_gpu_pq.py -q -k "pq" --timeout=400 > $O/tests.txt 2>&1; tail -n 3 $O/tests.txt
timeout 100 python tools/fuzz_parity.py 40 9900 > $O/fuzz.txt 2>&1; tail -n 1 $O/fuzz.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
PROBE_PLAIN=0 PROBE_OUT=$O/probe.jsonl timeout 400 rocprofv3 --kernel-trace -f csv -d /tmp/kt -o kt -- python $R/tools/hnswpq_probe.py 10000000 64:32 1344,1408 0 > $O/trace.out 2> $O/trace.err
python3 - > $O/pq_kernels.txt 2>&1 <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "hnsw_pq" in n or "pq_lut" in n:
        agg[n[:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for n, v in agg.items():
    print(f"{n:92s} launches {len(v):3d}  last {v[-1]:9.3f} ms  max {max(v):9.3f} ms")
PY
cut -c1-200 $O/pq_kernels.txt; grep '"pq"' $O/trace.out | cut -c1-230
