#!/bin/bash
# round 4, GPU call H: short-row burst fix + latency kernel on rows8 — parity, the reference's published point (1 M x 128, ef 20), 10 M latency
mkdir -p gpurun_out/r04h
O=gpurun_out/r04h
timeout 900 python -m pytest tests/test_gpu_rows8.py tests/test_gpu_walk2.py tests/test_gpu_hnsw.py tests/test_gpu_round2.py -m gpu -q -x --timeout=600 > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/tests.txt; tail -4 $O/tests.txt
EV8_AB_DIM=128 EV8_AB_CFG_EF=20 timeout 600 python tools/ev8_ab.py 1000000 0 uniform 20,64 > $O/ev8_h1.json 2> $O/ev8_h1.err; tail -c 1600 $O/ev8_h1.json; echo
EV8_AB_DIM=256 EV8_AB_CFG_EF=64 timeout 600 python tools/ev8_ab.py 2000000 1 normal 64,128 > $O/ev8_256f16.json 2> $O/ev8_256f16.err; tail -c 1600 $O/ev8_256f16.json; echo
timeout 600 python tools/latency.py 10000000 0 > $O/latency_rows8.txt 2>&1
COLTT_ROWS8=0 timeout 600 python tools/latency.py 10000000 0 > $O/latency_norows8.txt 2>&1
echo "--- with rows8"; grep "latency kernel" $O/latency_rows8.txt | cut -c1-160; echo "--- without"; grep "latency kernel" $O/latency_norows8.txt | cut -c1-160
