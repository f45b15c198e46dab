#!/bin/bash
# round 4, GPU call L: binary16 Commit / Load test; bit-map visited set A/B on one index per shape
mkdir -p gpurun_out/r04l
O=gpurun_out/r04l
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_hnsw.py -x -q -m gpu > $O/tests.txt 2>&1; tail -n 4 $O/tests.txt
timeout 600 python tools/knob_sweep.py 10000000 1 lowrank:32:1.0 1024 - COLTT_VISBITS=1 COLTT_VISBITS=1,COLTT_WALK2=6 COLTT_WALK2=6 COLTT_VISBITS=1,COLTT_BLOOM_KB=4 > $O/sweep_f16_ef1024.json 2> $O/sweep_f16_ef1024.err
cat $O/sweep_f16_ef1024.err | cut -c1-200
timeout 600 python tools/knob_sweep.py 10000000 0 normal 256 - COLTT_VISBITS=1 COLTT_VISBITS=1,COLTT_WALK2=6 > $O/sweep_f32_ef256.json 2> $O/sweep_f32_ef256.err
cat $O/sweep_f32_ef256.err | cut -c1-200
