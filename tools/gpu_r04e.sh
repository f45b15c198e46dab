#!/bin/bash
# round 4, GPU call E: eight-lane core without load predication, per-family rows x burst; C3 power / clock trace at three tile shapes
mkdir -p gpurun_out/r04e
O=gpurun_out/r04e
timeout 900 python -m pytest tests/test_gpu_rows8.py tests/test_gpu_walk2.py -m gpu -q -x --timeout=600 > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/tests.txt; tail -4 $O/tests.txt
for v in default allr2u6 allr1u12 r2u4; do
  if [ $v = default ]; then unset COLTT_LIB; else export COLTT_LIB=$PWD/coltt_amd/variants/libcoltt_$v.so; fi
  timeout 600 python tools/ev8_ab.py 10000000 0 normal 128,256 > $O/ev8_f32_$v.json 2> $O/ev8_f32_$v.err
  timeout 600 python tools/ev8_ab.py 10000000 1 lowrank:32:1.0 128,256,1024 > $O/ev8_f16_$v.json 2> $O/ev8_f16_$v.err
  python - <<P
import json
for t in ("f32", "f16"):
    try:
        d = json.load(open("$O/ev8_%s_$v.json" % t))
        print("$v", t, {ef: (round(r["eight_lanes"]["ms_per_launch"], 2), round(r["lane_pairs"]["ms_per_launch"], 2), round(r["speedup"], 3), round(r["eight_lanes"]["frac_of_hbm_peak"], 3), r["identical"]) for ef, r in d["ef"].items()})
    except Exception as e:
        print("$v", t, "failed", e)
P
done
unset COLTT_LIB
for v in default bm384 nsa4; do
  if [ $v = default ]; then unset COLTT_LIB; else export COLTT_LIB=$PWD/coltt_amd/variants/libcoltt_$v.so; fi
  POWER_PROBE_RAW=$O/power_c3_$v.raw.json timeout 300 python tools/power_probe.py 10000000,768,1,256 > $O/power_c3_$v.txt 2>&1
  tail -n 1 $O/power_c3_$v.txt | cut -c1-400
done
