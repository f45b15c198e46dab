#!/bin/bash
# round 5, GPU call R: the whole GPU suite + smoke on the library with the split re-rank, then a quantiser-shape sweep at smaller ef
mkdir -p gpurun_out/r05r
O=$PWD/gpurun_out/r05r
timeout 1200 python -m pytest tests -m gpu -q --timeout=900 > $O/suite.txt 2>&1
echo "suite rc=$?" >> $O/suite.txt; grep -n "passed\|failed" $O/suite.txt | tail -n 2; grep -n "^FAILED" $O/suite.txt | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
PROBE_OUT=$O/probe.jsonl timeout 600 python tools/hnswpq_probe.py 10000000 64:32,96:32,128:32,128:16,96:64 1024,1152,1280,1408 0 > $O/probe.out 2> $O/probe.err
echo "probe rc=$?"; grep '"pq"' $O/probe.out | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['m'],r['C'],r['ef'],r['recall'],r['qps'])"
grep '"plain"' $O/probe.out | cut -c1-120
