#!/bin/bash
# round 4, GPU call G: f8 rows through the matrix cores (new), then the whole GPU suite + smoke, then the driver's bench command
mkdir -p gpurun_out/r04g
O=gpurun_out/r04g
timeout 600 python -m pytest tests/test_gpu_round4.py -m gpu -q -x --timeout=600 > $O/round4_tests.txt 2>&1
echo "round4 tests rc=$?" >> $O/round4_tests.txt; tail -12 $O/round4_tests.txt
timeout 1200 python -m pytest tests -m gpu -q --timeout=900 --deselect tests/test_gpu_round4.py > $O/suite.txt 2>&1
echo "suite rc=$?" >> $O/suite.txt; tail -6 $O/suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.out 2> $O/bench.err
echo "bench rc=$?"; tail -n 1 $O/bench.out | cut -c1-3500; tail -4 $O/bench.err
cp bench_full.json $O/ 2>/dev/null
