#!/bin/bash
# round 5, GPU call Q: the exact re-rank as two kernels of its own (walk kernel without rows): parity, randomised parity, then the operating-point leg
mkdir -p gpurun_out/r05q
O=$PWD/gpurun_out/r05q
timeout 600 python -m pytest tests/test_gpu_round5.py -m gpu -q --timeout=500 > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -n "passed\|failed" $O/tests.txt | tail -n 2; grep -n "^FAILED" $O/tests.txt | head
timeout 100 python tools/fuzz_parity.py 70 8800 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -n 2 $O/fuzz.txt | cut -c1-400
( time python bench.py --gpus 1 --steps 20 --warmup 5 --legs op --no-cpu-baseline ) > $O/bench_op.out 2> $O/bench_op.err
echo "bench rc=$?"; tail -n 1 $O/bench_op.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['op']))"; cp bench_full.json $O/bench_full_op.json; tail -n 3 $O/bench_op.err
python -c "
import json; d=json.load(open('$O/bench_full_op.json')); pw=d['operating_point']['pq_walk']; print(pw['qps_vs_ef']); print(d['operating_point']['qps_vs_ef'])"
