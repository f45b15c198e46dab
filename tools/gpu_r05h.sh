#!/bin/bash
mkdir -p gpurun_out/r05h
O=$PWD/gpurun_out/r05h
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_rows8.py -m gpu -q --timeout=600 > $O/tests.txt 2>&1
echo "tests rc=$?"; grep -n "passed\|failed" $O/tests.txt | tail -n 3; grep -n "^FAILED" $O/tests.txt | head
PROBE_OUT=$O/probe.jsonl timeout 800 python tools/hnswpq_probe.py 10000000 64:16,64:32,64:64,96:16,96:32,48:64 1536,1792,2048 0 > $O/probe.out 2> $O/probe.err
echo "probe rc=$?"; cat $O/probe.out | cut -c1-215; tail -n 3 $O/probe.err
