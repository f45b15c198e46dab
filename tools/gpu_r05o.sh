#!/bin/bash
mkdir -p gpurun_out/r05o
O=$PWD/gpurun_out/r05o
timeout 600 python -m pytest tests/test_gpu_group.py tests/test_bench_line.py -m gpu -q -s --timeout=500 > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -n "passed\|failed" $O/tests.txt | tail -n 2; grep -n "group pipeline" $O/tests.txt
( time timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --legs g8 --no-cpu-baseline ) > $O/g8.out 2> $O/g8.err
echo "g8 rc=$?"; tail -n 1 $O/g8.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d.get('g8')))"; tail -n 3 $O/g8.err
