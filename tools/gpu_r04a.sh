#!/bin/bash
# round 4, GPU call A: new tests first, then the whole GPU suite, then the default bench run (line size, wall time, PQ leg)
mkdir -p gpurun_out/r04a
python -m pytest tests/test_gpu_pq.py tests/test_gpu_round4.py tests/test_bench_line.py -m gpu -q -x --timeout=900 > gpurun_out/r04a/new_tests.txt 2>&1
echo "new tests rc=$?" >> gpurun_out/r04a/new_tests.txt
tail -30 gpurun_out/r04a/new_tests.txt
python -m pytest tests -m gpu -q --timeout=900 --deselect tests/test_gpu_pq.py --deselect tests/test_gpu_round4.py --deselect tests/test_bench_line.py > gpurun_out/r04a/suite.txt 2>&1
echo "suite rc=$?" >> gpurun_out/r04a/suite.txt
tail -8 gpurun_out/r04a/suite.txt
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r04a/bench.out 2> gpurun_out/r04a/bench.err
echo "bench rc=$?"
tail -c 3000 gpurun_out/r04a/bench.out; tail -5 gpurun_out/r04a/bench.err
cp bench_full.json gpurun_out/r04a/ 2>/dev/null
