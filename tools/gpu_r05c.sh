#!/bin/bash
# round 5, GPU call C: binary16 tables + code-row prefetch in the product-quantised walk: parity tests again, then the 10 M probe over more shapes;
# the streamed group search with its timing printed
mkdir -p gpurun_out/r05c
O=$PWD/gpurun_out/r05c
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_pq.py "tests/test_gpu_group.py::test_streamed_shard_search_overlaps_exchange_and_merge_and_answers_identically" -m gpu -q -s --timeout=600 > $O/new_tests.txt 2>&1
echo "new tests rc=$?"; grep -n "passed\|failed" $O/new_tests.txt | tail -n 3; grep -n "group pipeline" $O/new_tests.txt | tail -n 2
PROBE_OUT=$O/hnswpq_probe.jsonl timeout 900 python tools/hnswpq_probe.py 10000000 32,48,64,96 1024,1536,2048 0,128 > $O/probe.out 2> $O/probe.err
echo "probe rc=$?"; cat $O/probe.out | cut -c1-260; tail -n 3 $O/probe.err
