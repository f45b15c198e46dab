#!/bin/bash
# round 5, GPU call F: ONE row array (line-transposed rows are the index's only copy; R8 variants of the pair-owned kernels, builder, read-backs):
# the whole GPU suite, smoke, the randomised parity run, then the op-point A/B against the previous layout's numbers (probe, plain walk only)
mkdir -p gpurun_out/r05f
O=$PWD/gpurun_out/r05f
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/suite.txt 2>&1
echo "suite rc=$?" >> $O/suite.txt; grep -n "passed\|failed" $O/suite.txt | tail -n 3; grep -n "^FAILED\|^ERROR" $O/suite.txt | head -n 12
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
timeout 130 python tools/fuzz_parity.py 100 5005 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -n 2 $O/fuzz.txt | cut -c1-300
PROBE_OUT=$O/probe.jsonl timeout 600 python tools/hnswpq_probe.py 10000000 32 1024 0 > $O/probe.out 2> $O/probe.err
echo "probe rc=$?"; cat $O/probe.out | cut -c1-250; tail -n 3 $O/probe.err
