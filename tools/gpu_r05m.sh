#!/bin/bash
# round 5, GPU call M: A/B of the speculative requests in the product-quantised walk (-DCOLTT_PQ_SPEC=1 variant against the shipped library), same probe, one process each
mkdir -p gpurun_out/r05m
O=$PWD/gpurun_out/r05m
PROBE_OUT=$O/probe_default.jsonl timeout 500 python tools/hnswpq_probe.py 10000000 64:32 1280,1408,1536,2048 0 > $O/probe_default.out 2> $O/probe_default.err; cut -c1-200 $O/probe_default.out
COLTT_LIB=$PWD/coltt_amd/variants/libcoltt_pqspec.so PROBE_OUT=$O/probe_spec.jsonl timeout 500 python tools/hnswpq_probe.py 10000000 64:32 1280,1408,1536,2048 0 > $O/probe_spec.out 2> $O/probe_spec.err; cut -c1-200 $O/probe_spec.out
