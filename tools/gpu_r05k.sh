#!/bin/bash
# round 5, GPU call K: the operating-point walk (10 M x 768 f16, ef 1024) at 3 waves per SIMD: shallower eight-lane bursts (COLTT_G8_U_H16 = 6 / 4:
# 154 / 129 VGPRs instead of 226) x 8 / 10 / 12 resident waves per CU, one index per library
mkdir -p gpurun_out/r05k
O=$PWD/gpurun_out/r05k
timeout 400 python tools/knob_sweep.py 10000000 1 lowrank:32:1.0 1024 - COLTT_WAVES_PER_CU=10 COLTT_WAVES_PER_CU=12 > $O/sweep_default.json 2> $O/sweep_default.err; grep -c setting $O/sweep_default.err; cat $O/sweep_default.err | cut -c1-200
for V in h16u6 h16u4; do
  COLTT_LIB=$PWD/coltt_amd/variants/libcoltt_$V.so timeout 400 python tools/knob_sweep.py 10000000 1 lowrank:32:1.0 1024 - COLTT_WAVES_PER_CU=10 COLTT_WAVES_PER_CU=12 > $O/sweep_$V.json 2> $O/sweep_$V.err; echo $V; cat $O/sweep_$V.err | cut -c1-200
done
