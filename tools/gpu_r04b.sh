#!/bin/bash
# round 4, GPU call B: launcher test + short-row matrix-core tests, the two unmeasured round-3 flags, PQ probe + rocprof records
mkdir -p gpurun_out/r04b
O=gpurun_out/r04b
python -m pytest tests/test_bench_line.py tests/test_gpu_round4.py tests/test_gpu_pq.py "tests/test_gpu_flat.py::test_flat_mfma_many_tiles_per_workgroup_smallest_dim" -m gpu -q --timeout=900 > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/tests.txt; tail -15 $O/tests.txt
# C3 / C2 A/B of -DCOLTT_M2_PK_EPI (flat_mfma2.hpp:102)
for v in default pkepi; do
  if [ $v = default ]; then unset COLTT_LIB; else export COLTT_LIB=$PWD/coltt_amd/variants/libcoltt_$v.so; fi
  python tools/flat_ab.py 10000000,768,1,256 1000000,768,0,64 >> $O/flat_ab.txt 2>&1
done
unset COLTT_LIB
cat $O/flat_ab.txt
# single-query latency A/B of -DCOLTT_LAT_EVAL_PIPE (hnsw_lat.hpp:116)
python tools/latency.py 10000000 0 > $O/latency_default.txt 2>&1
COLTT_LIB=$PWD/coltt_amd/variants/libcoltt_evalpipe.so python tools/latency.py 10000000 0 > $O/latency_evalpipe.txt 2>&1
grep "latency kernel" $O/latency_default.txt | head -3; echo ---; grep "latency kernel" $O/latency_evalpipe.txt | head -3
# PQ probe, plain and under the kernel trace
python tools/pq_probe.py > $O/pq_probe.json 2> $O/pq_probe.err; cat $O/pq_probe.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d /tmp/pqprof -o pq -- python $GRAFT_REPO_ROOT/tools/pq_probe.py > /tmp/pqprof.out 2>&1
cd $GRAFT_REPO_ROOT
cp /tmp/pqprof/*kernel_stats.csv $O/pq_kernel_stats.csv 2>/dev/null; cp /tmp/pqprof/*kernel_trace.csv $O/pq_kernel_trace.csv 2>/dev/null
ls /tmp/pqprof | head; head -12 $O/pq_kernel_stats.csv
