#!/bin/bash
# round 4, GPU call C: the eight-lanes-per-row distance core — parity first, then the 10 M sweeps (f32 ef 128 headline, f16 lowrank ef curve)
mkdir -p gpurun_out/r04c
O=gpurun_out/r04c
timeout 900 python -m pytest tests/test_gpu_rows8.py -m gpu -q -x --timeout=600 > $O/rows8_tests.txt 2>&1
echo "rows8 tests rc=$?" >> $O/rows8_tests.txt; tail -25 $O/rows8_tests.txt
timeout 900 python -m pytest tests/test_gpu_walk2.py tests/test_gpu_hnsw.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_group.py -m gpu -q --timeout=600 > $O/hnsw_suites.txt 2>&1
echo "hnsw suites rc=$?" >> $O/hnsw_suites.txt; tail -6 $O/hnsw_suites.txt
# headline shape: 10 M x 768 f32, ef 128 (LDS-visited walk) and ef 256 / 1024, eight-lane core vs pair-owned rows in ONE process
timeout 1200 python tools/ev8_ab.py 10000000 0 normal 128,256,1024 > $O/ev8_f32.json 2> $O/ev8_f32.err; tail -c 1500 $O/ev8_f32.json
# operating-point shape: 10 M x 768 f16 codes, lowrank:32
timeout 1200 python tools/ev8_ab.py 10000000 1 lowrank:32:1.0 128,256,512,1024 > $O/ev8_f16.json 2> $O/ev8_f16.err; tail -c 1800 $O/ev8_f16.json
