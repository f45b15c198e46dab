#!/bin/bash
# round 5, GPU call S: the table walk with compile-time table rows (immediate LDS offsets, SDWA byte extraction, v_fma_mix_f32) — parity, then A/B on one box:
# old library | new | new + runner-up adjacency at pop time; Bloom filter / resident-wave knobs swept in-process
mkdir -p gpurun_out/r05s
O=$PWD/gpurun_out/r05s
timeout 600 python -m pytest tests/test_gpu_round5.py -q -k "pq" --timeout=500 > $O/tests.txt 2>&1; tail -n 3 $O/tests.txt
timeout 120 python tools/fuzz_parity.py 50 9100 > $O/fuzz.txt 2>&1; tail -n 2 $O/fuzz.txt
K='|COLTT_PQ_BLOOM_KB=0|COLTT_PQ_WAVES=16|COLTT_PQ_BLOOM_KB=0;COLTT_PQ_WAVES=16'
PROBE_KNOBS="$K" PROBE_OUT=$O/new.jsonl timeout 400 python tools/hnswpq_probe.py 10000000 64:32 1024,1408 0 > $O/new.out 2> $O/new.err
COLTT_LIB=$PWD/coltt_amd/variants/libcoltt_old.so PROBE_PLAIN=0 PROBE_OUT=$O/old.jsonl timeout 400 python tools/hnswpq_probe.py 10000000 64:32 1024,1408 0 > $O/old.out 2> $O/old.err
COLTT_LIB=$PWD/coltt_amd/variants/libcoltt_radj.so PROBE_PLAIN=0 PROBE_KNOBS='|COLTT_PQ_BLOOM_KB=0;COLTT_PQ_WAVES=16' PROBE_OUT=$O/radj.jsonl timeout 400 python tools/hnswpq_probe.py 10000000 64:32 1024,1408 0 > $O/radj.out 2> $O/radj.err
for f in old new radj; do echo "== $f"; python - $O/$f.jsonl <<'PY'
import sys, json
for l in open(sys.argv[1]):
    r = json.loads(l)
    if r.get("kind") == "pq": print(r.get("knobs"), r.get("ef"), r.get("recall"), r.get("qps"), r.get("kernel_ms"), r.get("error", ""))
    else: print("plain", r["ef"], r["recall"], r["qps"])
PY
done
