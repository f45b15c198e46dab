#!/bin/bash
# round 5, GPU call Y: the product-quantised search as query groups pipelined over 2 / 3 streams (COLTT_PQ_STREAMS) — parity, then A/B on one index
mkdir -p gpurun_out/r05y
O=$PWD/gpurun_out/r05y
timeout 400 python -m pytest tests/test_gpu_round5.py -q -k "pq" --timeout=300 > $O/tests.txt 2>&1; tail -n 4 $O/tests.txt
PROBE_KNOBS='|COLTT_PQ_STREAMS=3|COLTT_PQ_STREAMS=2|COLTT_PQ_STREAMS=1|COLTT_PQ_STREAMS=3' PROBE_PLAIN=0 PROBE_OUT=$O/ab.jsonl timeout 400 python tools/hnswpq_probe.py 10000000 64:32 1024,1344,1408 0 > $O/ab.out 2> $O/ab.err
echo "probe rc=$?"; python - $O/ab.jsonl <<'PY'
import sys, json
for l in open(sys.argv[1]):
    r = json.loads(l)
    if r.get("kind") == "pq": print(r.get("knobs"), r.get("ef"), r.get("recall"), r.get("qps"), r.get("kernel_ms"), r.get("n_dist"), r.get("error", ""))
PY
tail -n 3 $O/ab.err
