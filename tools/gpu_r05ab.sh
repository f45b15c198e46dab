#!/bin/bash
# round 5, GPU call AB: the whole GPU suite + smoke on the last library (binary16 tables written by pq_lut16_kernel)
mkdir -p gpurun_out/r05ab
O=$PWD/gpurun_out/r05ab
timeout 1000 python -m pytest tests -m gpu -q --timeout=900 > $O/suite.txt 2>&1
echo "suite rc=$?" >> $O/suite.txt; grep -n "passed\|failed" $O/suite.txt | tail -n 2; grep -n "^FAILED" $O/suite.txt | head
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
