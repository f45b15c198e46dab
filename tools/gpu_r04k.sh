#!/bin/bash
# round 4, GPU call K: resident waves / Bloom size of the HBM-visited walk with the eight-lane core, one index per shape
mkdir -p gpurun_out/r04k
O=gpurun_out/r04k
timeout 600 python tools/knob_sweep.py 10000000 1 lowrank:32:1.0 1024 - COLTT_WAVES_PER_CU=6 COLTT_WAVES_PER_CU=4 COLTT_WAVES_PER_CU=6,COLTT_BLOOM_KB=16 COLTT_WAVES_PER_CU=4,COLTT_BLOOM_KB=32 COLTT_BLOOM_KB=4 COLTT_EV8=0 COLTT_EV8=0,COLTT_WAVES_PER_CU=6 > $O/sweep_f16_ef1024.json 2> $O/sweep_f16_ef1024.err
cat $O/sweep_f16_ef1024.err | cut -c1-200
timeout 600 python tools/knob_sweep.py 10000000 0 normal 256 - COLTT_WAVES_PER_CU=3 COLTT_WAVES_PER_CU=6 COLTT_WAVES_PER_CU=8 COLTT_EV8=0 > $O/sweep_f32_ef256.json 2> $O/sweep_f32_ef256.err
cat $O/sweep_f32_ef256.err | cut -c1-200
