#!/bin/bash
# round 5, GPU call A: (1) batched builder vs the reference's sequential Insert: recall / n_dist per ef on the same 300 k x 768 f16
# lowrank:32:1.0 collection; (2) library f16 GEMM of the C3 shape under the power probe, beside the shipped kernel on the same box
mkdir -p gpurun_out/r05a
O=$PWD/gpurun_out/r05a
( timeout 240 python tools/c3_yardstick.py > $O/yardstick.jsonl 2> $O/yardstick.err; echo "yardstick rc=$?"; cat $O/yardstick.jsonl | cut -c1-400 )
( POWER_PROBE_RAW=$O/power_default.raw.json timeout 200 python tools/power_probe.py > $O/power_default.txt 2>&1; echo "probe rc=$?"; tail -n 1 $O/power_default.txt | cut -c1-600 )
BQ_OUT=$O/build_quality.jsonl timeout 1000 python tools/build_quality.py 300000 1 lowrank:32:1.0 bench,1024,1 > $O/bq.out 2> $O/bq.err
echo "bq rc=$?"; cat $O/bq.out | cut -c1-1200; tail -n 3 $O/bq.err
