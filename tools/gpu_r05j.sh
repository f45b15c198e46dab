#!/bin/bash
mkdir -p gpurun_out/r05j
O=$PWD/gpurun_out/r05j
( time python bench.py --gpus 1 --steps 20 --warmup 5 --legs op --no-cpu-baseline ) > $O/bench_op.out 2> $O/bench_op.err
echo "bench rc=$?"; tail -n 1 $O/bench_op.out | cut -c1-2000; tail -n 3 $O/bench_op.err
timeout 700 python -m pytest tests -m gpu -q --timeout=600 --durations=12 -x > $O/suite.txt 2>&1; grep -n "passed\|failed" $O/suite.txt | tail -n 2; grep -n "s call\|s setup" $O/suite.txt | head -n 14
timeout 100 python tools/fuzz_parity.py 70 6100 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -n 2 $O/fuzz.txt | cut -c1-300
