#!/bin/bash
# round 5, GPU call I: the library after the one-row-array and product-quantised-walk work — whole GPU suite + smoke, the driver's bench command
# plain, under the kernel trace, and the PMC passes (HNSW + PQ legs of bench.py; the FLAT C2 / C3 shapes through tools/flat_ab.py)
mkdir -p gpurun_out/r05i
O=$PWD/gpurun_out/r05i
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/suite.txt 2>&1
echo "suite rc=$?" >> $O/suite.txt; grep -n "passed\|failed" $O/suite.txt | tail -n 2; grep -n "^FAILED" $O/suite.txt | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.out 2> $O/bench.err
echo "bench rc=$?"; tail -n 1 $O/bench.out | cut -c1-1700; tail -n 4 $O/bench.err; cp bench_full.json $O/bench_full.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d /tmp/kt -o kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_under_rocprof.out 2> $O/bench_under_rocprof.err
cp $R/bench_full.json $O/bench_full_under_rocprof.json
cp /tmp/kt/*kernel_stats.csv $O/kernel_stats.csv; python $R/tools/trace_by_grid.py /tmp/kt/*kernel_trace.csv 1.0 > $O/kernel_stats_by_grid.csv; head -n 6 $O/kernel_stats_by_grid.csv | cut -c1-220
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "hnsw_search|flat_scan_kernel|pq_scan_kernel" -f csv -d /tmp/pmc -o p -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --legs op,pq > $O/bench_under_pmc.out 2> $O/bench_under_pmc.err
cp $R/bench_full.json $O/bench_full_under_pmc.json
cp /tmp/pmc/*counter_collection.csv $O/pmc_fetch_size_raw.csv
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "mfma|flat_pick|flat_rescore|flat_select" -f csv -d /tmp/pmcf -o p -- python $R/tools/flat_ab.py 1000000,768,0,64 10000000,768,1,256 > $O/flat_under_pmc.out 2> $O/flat_under_pmc.err
cp /tmp/pmcf/*counter_collection.csv $O/pmc_flat_fetch_size_raw.csv; ls -la $O | head -n 20
cd $R
