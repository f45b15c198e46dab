#!/bin/bash
# round 5, GPU call T: one-instruction DPP reduction steps (v_min_u32_dpp / v_max_u32_dpp) in the delta-set walks — parity, then previous library vs new on one box
mkdir -p gpurun_out/r05t
O=$PWD/gpurun_out/r05t
timeout 900 python -m pytest tests/test_gpu_walk2.py tests/test_gpu_round5.py tests/test_gpu_rows8.py -q --timeout=600 > $O/tests.txt 2>&1; tail -n 3 $O/tests.txt
timeout 120 python tools/fuzz_parity.py 45 9400 > $O/fuzz.txt 2>&1; tail -n 1 $O/fuzz.txt
PROBE_OUT=$O/new.jsonl timeout 400 python tools/hnswpq_probe.py 10000000 64:32 1024,1280,1408 0 > $O/new.out 2> $O/new.err
COLTT_LIB=$PWD/coltt_amd/variants/libcoltt_prev.so PROBE_OUT=$O/prev.jsonl timeout 400 python tools/hnswpq_probe.py 10000000 64:32 1024,1280,1408 0 > $O/prev.out 2> $O/prev.err
for f in prev new; do echo "== $f"; python - $O/$f.jsonl <<'PY'
import sys, json
for l in open(sys.argv[1]):
    r = json.loads(l)
    if r.get("kind") == "pq": print("pq", r.get("ef"), r.get("recall"), r.get("qps"), r.get("kernel_ms"), r.get("error", ""))
    else: print("plain", r["ef"], r["recall"], r["qps"], r["kernel_ms"])
PY
done
