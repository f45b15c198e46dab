#!/bin/bash
# round 4, GPU call F: index arrays reserved once (coltt_hnsw_reserve) — the eight-lane A/B again on both 10 M shapes, reserved vs grown
mkdir -p gpurun_out/r04f
O=gpurun_out/r04f
timeout 600 python -m pytest tests/test_gpu_rows8.py -m gpu -q -x --timeout=600 > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/tests.txt; tail -3 $O/tests.txt
for mode in reserve noreserve reserve; do
  tag=$mode; [ -f $O/ev8_f16_$tag.json ] && tag=${mode}2
  timeout 600 python tools/ev8_ab.py 10000000 1 lowrank:32:1.0 128,256,1024 $mode > $O/ev8_f16_$tag.json 2> $O/ev8_f16_$tag.err
  timeout 600 python tools/ev8_ab.py 10000000 0 normal 128,256,1024 $mode > $O/ev8_f32_$tag.json 2> $O/ev8_f32_$tag.err
  python - <<P
import json
for t in ("f16", "f32"):
    try:
        d = json.load(open("$O/ev8_%s_$tag.json" % t))
        print("$tag", t, "build", round(d["build_s"], 1), {ef: (round(r["eight_lanes"]["ms_per_launch"], 2), round(r["lane_pairs"]["ms_per_launch"], 2), round(r["speedup"], 3), round(r["eight_lanes"]["frac_of_hbm_peak"], 3), round(r["lane_pairs"]["frac_of_hbm_peak"], 3), r["identical"]) for ef, r in d["ef"].items()})
    except Exception as e:
        print("$tag", t, "failed", e)
P
done
