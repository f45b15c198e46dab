#!/bin/bash
# round 5, GPU call E: table rows as long as the centroid count (16-centroid quantisers: 2-4 KiB tables): parity, then the 10 M probe
mkdir -p gpurun_out/r05e
O=$PWD/gpurun_out/r05e
timeout 900 python -m pytest tests/test_gpu_round5.py -m gpu -q --timeout=600 > $O/new_tests.txt 2>&1
echo "new tests rc=$?"; grep -n "passed\|failed" $O/new_tests.txt | tail -n 3
PROBE_OUT=$O/hnswpq_probe.jsonl timeout 900 python tools/hnswpq_probe.py 10000000 32,64:16,96:16,128:16,48:64 1024,1280,1536,2048 0 > $O/probe.out 2> $O/probe.err
echo "probe rc=$?"; cat $O/probe.out | cut -c1-250; tail -n 3 $O/probe.err
