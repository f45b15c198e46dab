#!/bin/bash
# round 5, GPU call N: the g8 leg (8 members on one device, streamed against serial) on its own, then the driver's command with it inside
mkdir -p gpurun_out/r05n
O=$PWD/gpurun_out/r05n
( time timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --n 300000 --legs g8 --no-cpu-baseline ) > $O/g8_small.out 2> $O/g8_small.err
echo "small rc=$?"; tail -n 1 $O/g8_small.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d.get('g8')))"; tail -n 4 $O/g8_small.err
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.out 2> $O/bench.err
echo "bench rc=$?"; tail -n 1 $O/bench.out | wc -c; tail -n 1 $O/bench.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d.get(k) for k in ('value','ms_per_step','roofline','op','g8','wall_s')}))"; tail -n 4 $O/bench.err; cp bench_full.json $O/bench_full.json
