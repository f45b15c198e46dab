#!/bin/bash
# round 5, GPU call D: bench.py with the product-quantised walk inside the operating-point leg and the full-size oracle samples of C3 / c3f8:
# a small run first (plumbing), then the driver's command
mkdir -p gpurun_out/r05d
O=$PWD/gpurun_out/r05d
( time timeout 600 python bench.py --n 300000 --queries 2000 --steps 3 --warmup 1 --legs op,c3 --cpu-seconds 1 ) > $O/small.out 2> $O/small.err
echo "small rc=$?"; tail -n 1 $O/small.out | cut -c1-1500; tail -n 5 $O/small.err
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.out 2> $O/bench.err
echo "bench rc=$?"; tail -n 1 $O/bench.out | cut -c1-3800; tail -n 4 $O/bench.err; cp bench_full.json $O/bench_full.json
