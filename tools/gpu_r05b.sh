#!/bin/bash
# round 5, GPU call B: new tests first (HNSW over PQ codes == oracle, pipelined group exchange, rows8 fallback, build quality), the whole
# GPU suite, then the product-quantised walk at the operating-point shape (10 M x 768 f16 lowrank:32:1.0)
mkdir -p gpurun_out/r05b
O=$PWD/gpurun_out/r05b
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_group.py -m gpu -q -x -s --timeout=600 > $O/new_tests.txt 2>&1
echo "new tests rc=$?"; grep -n "passed\|failed\|Error\|error" $O/new_tests.txt | tail -n 8; grep -n "group pipeline\|build quality" $O/new_tests.txt | tail -n 8
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/suite.txt 2>&1
echo "suite rc=$?" >> $O/suite.txt; grep -n "passed\|failed" $O/suite.txt | tail -n 3
PROBE_OUT=$O/hnswpq_probe.jsonl timeout 900 python tools/hnswpq_probe.py 10000000 32,96 512,1024,2048 0,128 > $O/probe.out 2> $O/probe.err
echo "probe rc=$?"; cat $O/probe.out | cut -c1-330; tail -n 3 $O/probe.err
