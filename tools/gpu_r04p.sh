#!/bin/bash
# round 4, GPU call P: two-pass epilogue with the unrolled element test — FLAT tests; same-box A/B against the previous library
# (coltt_amd/libcoltt_gpu_prev.so = HEAD before the epilogue change, built from a copy of the tree) under the segment schedules
mkdir -p gpurun_out/r04p
O=gpurun_out/r04p
timeout 900 python -m pytest tests/test_gpu_flat.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_cflat.py tests/test_gpu_cpp_mirror.py tests/test_gpu_group.py -x -q -m gpu > $O/tests.txt 2>&1; tail -n 3 $O/tests.txt
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 400 python bench.py --gpus 1 --steps 3 --warmup 1 --n 200000 --no-cpu-baseline --legs c2,c3,f3 > $O/bench_$name.out 2> $O/bench_$name.err
  python - "$O/bench_$name.out" "$name" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-2])
s = d["secondary"]
c2 = s["c2"]; c3 = s["c3"]; f3 = s["f3"]["lists"]
out = {"c2_kernels_ms": round(c2["ms_per_batch_kernels"], 4), "c2_same": c2["identical_to_exact_mode"], "c3_kernels_ms": round(c3["ms_per_batch_kernels"], 4), "c3_same": c3["identical_to_exact_mode"]}
for ln, l in f3.items():
    for k, v in l.items():
        if isinstance(v, dict) and "mfma" in k:
            out[f"{l['candidates']}_{k}"] = (round(v["kernels_ms"], 4), v.get("equals_exact_mode"))
print(sys.argv[2], json.dumps(out))
P
}
PREV=$PWD/coltt_amd/libcoltt_gpu_prev.so
run prev_default COLTT_LIB=$PREV
run new_default COLTT_X=1
run prev_two_8k COLTT_LIB=$PREV COLTT_MFMA_GROW=100000 COLTT_MFMA_SEED=8192
run new_two_4k COLTT_MFMA_GROW=100000 COLTT_MFMA_SEED=4096
run new_two_8k COLTT_MFMA_GROW=100000 COLTT_MFMA_SEED=8192
run new_two_16k COLTT_MFMA_GROW=100000 COLTT_MFMA_SEED=16384
run new_three_g64 COLTT_MFMA_GROW=64 COLTT_MFMA_SEED=4096
run new_three_g32_s2k COLTT_MFMA_GROW=32 COLTT_MFMA_SEED=2048
run prev_default_again COLTT_LIB=$PREV
run new_default_again COLTT_X=1
