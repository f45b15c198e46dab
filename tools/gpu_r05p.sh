#!/bin/bash
mkdir -p gpurun_out/r05p
O=$PWD/gpurun_out/r05p
COLTT_LIB=$PWD/coltt_amd/variants/libcoltt_phase.so timeout 500 python tools/hnswpq_probe.py 10000000 64:32,32:256 1024,1408 0 > $O/probe.out 2> $O/probe.err; grep "phase" $O/probe.err | cut -c1-330
