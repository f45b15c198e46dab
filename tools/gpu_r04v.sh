#!/bin/bash
# round 4, GPU call V: last library (PQ host round trip merged) — whole GPU suite + smoke; the PQ leg alone
mkdir -p gpurun_out/r04v
O=$PWD/gpurun_out/r04v
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/suite.txt 2>&1
echo "suite rc=$?" >> $O/suite.txt; grep -n "passed\|failed" $O/suite.txt | tail -n 2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
timeout 400 python bench.py --gpus 1 --steps 3 --warmup 1 --n 200000 --legs pq > $O/bench_pq.out 2> $O/bench_pq.err
tail -n 1 $O/bench_pq.out | cut -c1-200; python - <<'P'
import json
d = json.loads(open("gpurun_out/r04v/bench_pq.out").read().strip().splitlines()[-2])
p = d["secondary"]["pq"]; print(json.dumps({k: p[k] for k in ("value", "ms_per_batch_kernels", "single_query_scan_launch_ms", "batch_64_queries_per_s")}), p["roofline"]["frac"], p.get("cpu_baseline", {}).get("gpu_equals_oracle_on_sample"))
P
