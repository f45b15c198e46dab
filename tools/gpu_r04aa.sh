#!/bin/bash
# round 4, GPU call AA: the last library (pick out of registers) — whole GPU suite + smoke + the driver's bench command
mkdir -p gpurun_out/r04aa
O=$PWD/gpurun_out/r04aa
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/suite.txt 2>&1
echo "suite rc=$?" >> $O/suite.txt; grep -n "passed\|failed" $O/suite.txt | tail -n 2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.out 2> $O/bench.err
echo "bench rc=$?"; tail -n 1 $O/bench.out | cut -c1-200; tail -n 4 $O/bench.err; cp bench_full.json $O/bench_full.json
