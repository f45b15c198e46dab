#!/bin/bash
# round 4, GPU call T: exact scan with the reservations of a group issued together — FLAT tests + same-box A/B (exact-mode legs of f3 / c2)
mkdir -p gpurun_out/r04t
O=gpurun_out/r04t
timeout 900 python -m pytest tests/test_gpu_flat.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_cflat.py -x -q -m gpu > $O/tests.txt 2>&1; grep -n "passed\|failed" $O/tests.txt | tail -n 2
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 400 python bench.py --gpus 1 --steps 3 --warmup 1 --n 200000 --no-cpu-baseline --legs c2,f3 > $O/bench_$name.out 2> $O/bench_$name.err
  python - "$O/bench_$name.out" "$name" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-2])
s = d["secondary"]
c2 = s["c2"]; f3 = s["f3"]["lists"]
out = {"c2_exact_ms": round(c2["exact_mode_ms_per_batch"], 4)}
for ln, l in f3.items():
    for k, v in l.items():
        if isinstance(v, dict) and "exact" in k and "chain" not in k:
            out[f"{l['candidates']}_{k}"] = (round(v["kernels_ms"], 4), v.get("equals_exact_mode"))
print(sys.argv[2], json.dumps(out))
P
}
PREV=$PWD/coltt_amd/libcoltt_gpu_prev.so
run prev COLTT_LIB=$PREV
run new COLTT_X=1
run prev_again COLTT_LIB=$PREV
run new_again COLTT_X=1
