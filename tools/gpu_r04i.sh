#!/bin/bash
# round 4, GPU call I: final library — whole GPU suite + smoke, the driver's bench command plain, under the kernel trace, under the PMC pass
mkdir -p gpurun_out/r04i
O=$PWD/gpurun_out/r04i
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/suite.txt 2>&1
echo "suite rc=$?" >> $O/suite.txt; tail -5 $O/suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.out 2> $O/bench.err
echo "bench rc=$?"; tail -n 1 $O/bench.out | cut -c1-1200; tail -4 $O/bench.err; cp bench_full.json $O/bench_full.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d /tmp/kt -o kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_under_rocprof.out 2> $O/bench_under_rocprof.err
cp $R/bench_full.json $O/bench_full_under_rocprof.json
ls /tmp/kt | head; cp /tmp/kt/*kernel_stats.csv $O/kernel_stats.csv; python $R/tools/trace_by_grid.py /tmp/kt/*kernel_trace.csv 1.0 > $O/kernel_stats_by_grid.csv; head -5 $O/kernel_stats_by_grid.csv | cut -c1-200
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "hnsw_search|flat_scan_kernel|pq_scan_kernel" -f csv -d /tmp/pmc -o p -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --legs op,pq > $O/bench_under_pmc.out 2> $O/bench_under_pmc.err
cp $R/bench_full.json $O/bench_full_under_pmc.json
cp /tmp/pmc/*counter_collection.csv $O/pmc_fetch_size_raw.csv; ls -la $O | head -20
cd $R
