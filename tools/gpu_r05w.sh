#!/bin/bash
# round 5, GPU call W: tables of binary16 denormals through v_fma_mix_f32 (parity), and one query at a time: plain walk vs product-quantised walk
mkdir -p gpurun_out/r05w
O=$PWD/gpurun_out/r05w
timeout 400 python -m pytest tests/test_gpu_round5.py -q -k "pq" --timeout=300 > $O/tests.txt 2>&1; tail -n 3 $O/tests.txt
PROBE_OUT=$O/latency.jsonl timeout 400 python tools/pq_latency_probe.py 10000000 > $O/latency.out 2> $O/latency.err; echo "probe rc=$?"; cat $O/latency.jsonl; tail -n 3 $O/latency.err
