#!/bin/bash
# round 5, GPU call L: speculative requests in the product-quantised walk (visited bytes + code rows of the runner-up's neighbours one expansion ahead):
# parity (tests + randomised), then the operating-point leg
mkdir -p gpurun_out/r05l
O=$PWD/gpurun_out/r05l
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_walk2.py tests/test_gpu_rows8.py -m gpu -q --timeout=500 > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -n "passed\|failed" $O/tests.txt | tail -n 2; grep -n "^FAILED" $O/tests.txt | head
timeout 100 python tools/fuzz_parity.py 70 7300 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -n 2 $O/fuzz.txt | cut -c1-400
( time python bench.py --gpus 1 --steps 20 --warmup 5 --legs op --no-cpu-baseline ) > $O/bench_op.out 2> $O/bench_op.err
echo "bench rc=$?"; tail -n 1 $O/bench_op.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['op']))"; cp bench_full.json $O/bench_full_op.json; tail -n 3 $O/bench_op.err
