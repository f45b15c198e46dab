#!/bin/bash
# round 5, GPU call AC: headline + operating-point legs of bench.py on the shipped library (no CPU legs)
mkdir -p gpurun_out/r05ac
O=$PWD/gpurun_out/r05ac
timeout 125 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --legs op > $O/bench.out 2> $O/bench.err
echo "bench rc=$?"; tail -n 1 $O/bench.out | cut -c1-1800; cp bench_full.json $O/bench_full.json 2>/dev/null
