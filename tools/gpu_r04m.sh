#!/bin/bash
# round 4, GPU call M: the final library — whole GPU suite + smoke, the driver's bench command, LDS counters of the PQ scan
mkdir -p gpurun_out/r04m
O=$PWD/gpurun_out/r04m
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/suite.txt 2>&1
echo "suite rc=$?" >> $O/suite.txt; tail -4 $O/suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.out 2> $O/bench.err
echo "bench rc=$?"; tail -n 1 $O/bench.out | cut -c1-900; tail -n 4 $O/bench.err; cp bench_full.json $O/bench_full.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-include-regex "pq_scan" -f csv -d /tmp/pl -o p -- python $R/tools/pq_probe.py 10000000 768 96 > $O/pq_lds.out 2> $O/pq_lds.err
ls /tmp/pl | head -5; cp /tmp/pl/*counter_collection.csv $O/pq_lds_raw.csv 2>/dev/null; head -3 $O/pq_lds_raw.csv | cut -c1-300
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-include-regex "pq_scan" -f csv -d /tmp/pl2 -o p -- python $R/tools/pq_probe.py 10000000 768 96 > $O/pq_lds2.out 2> $O/pq_lds2.err
cp /tmp/pl2/*counter_collection.csv $O/pq_lds2_raw.csv 2>/dev/null; head -3 $O/pq_lds2_raw.csv | cut -c1-300
cd $R
