#!/bin/bash
# round 5, GPU call V: the last library (table walk with immediate LDS offsets / SDWA / v_fma_mix, no Bloom filter there, one-instruction DPP steps) —
# whole GPU suite + smoke, the driver's bench command, the same command's headline + operating-point legs under the kernel trace, randomised parity
mkdir -p gpurun_out/r05v
O=$PWD/gpurun_out/r05v
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/suite.txt 2>&1
echo "suite rc=$?" >> $O/suite.txt; grep -n "passed\|failed" $O/suite.txt | tail -n 2; grep -n "^FAILED" $O/suite.txt | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.out 2> $O/bench.err
echo "bench rc=$?"; tail -n 1 $O/bench.out | cut -c1-2600; tail -n 4 $O/bench.err; cp bench_full.json $O/bench_full.json
timeout 100 python tools/fuzz_parity.py 60 9700 > $O/fuzz.txt 2>&1; tail -n 1 $O/fuzz.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/kt -o kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --legs op > $O/bench_under_rocprof.out 2> $O/bench_under_rocprof.err
cp $R/bench_full.json $O/bench_full_under_rocprof.json
cp /tmp/kt/*kernel_stats.csv $O/kernel_stats.csv; python $R/tools/trace_by_grid.py /tmp/kt/*kernel_trace.csv 1.0 > $O/kernel_stats_by_grid.csv; head -n 8 $O/kernel_stats_by_grid.csv | cut -c1-220
cd $R
