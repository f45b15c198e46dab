#!/bin/bash
# round 4, GPU call N: Reserve test; the matrix-core chain's segment schedule on C2 / f3 (two segments instead of three)
mkdir -p gpurun_out/r04n
O=gpurun_out/r04n
timeout 300 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "reserve or binary16" > $O/tests.txt 2>&1; tail -n 3 $O/tests.txt
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 400 python bench.py --gpus 1 --steps 3 --warmup 1 --n 200000 --no-cpu-baseline --legs c2,f3 > $O/bench_$name.out 2> $O/bench_$name.err
  python - "$O/bench_$name.out" "$name" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-2])
s = d["secondary"]
c2 = s["c2"]; f3 = s["f3"]["lists"]
out = {"c2_kernels_ms": round(c2["ms_per_batch_kernels"], 4), "c2_same": c2["identical_to_exact_mode"]}
for ln, l in f3.items():
    for k, v in l.items():
        if isinstance(v, dict) and "mfma" in k:
            out[f"{l['candidates']}_{k}"] = (round(v["kernels_ms"], 4), v.get("equals_exact_mode"))
print(sys.argv[2], json.dumps(out))
P
}
run default COLTT_X=1
run two_4k COLTT_MFMA_GROW=100000 COLTT_MFMA_SEED=4096
run two_16k COLTT_MFMA_GROW=100000 COLTT_MFMA_SEED=16384
run two_64k COLTT_MFMA_GROW=100000 COLTT_MFMA_SEED=65536
run three_g64 COLTT_MFMA_GROW=64 COLTT_MFMA_SEED=4096
