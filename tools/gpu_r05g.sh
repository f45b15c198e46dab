#!/bin/bash
mkdir -p gpurun_out/r05g
O=$PWD/gpurun_out/r05g
timeout 200 python tools/r8_debug.py > $O/r8_debug.txt 2>&1; cat $O/r8_debug.txt | cut -c1-200
PROBE_OUT=$O/probe.jsonl timeout 600 python tools/hnswpq_probe.py 10000000 32,64:16,128:16 1280,1536 0 > $O/probe.out 2> $O/probe.err
echo "probe rc=$?"; cat $O/probe.out | cut -c1-250; tail -n 3 $O/probe.err
