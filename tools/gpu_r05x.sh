#!/bin/bash
# round 5, GPU call X: counters of the product-quantised walk kernel (instruction mix, busy / wait cycles, HBM bytes), three separate PMC passes
mkdir -p gpurun_out/r05x
O=$PWD/gpurun_out/r05x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  name=$1; shift
  PROBE_PLAIN=0 timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "hnsw_pq_search_kernel" -f csv -d /tmp/pmc_$name -o p -- python $R/tools/hnswpq_probe.py 10000000 64:32 1408 0 > $O/$name.out 2> $O/$name.err
  echo "$name rc=$?"
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -n 1)
  [ -n "$f" ] && cp $f $O/pmc_$name.csv && python3 - $f <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows: agg[(r["Kernel_Name"][:60], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (kn, g, c), v in sorted(agg.items()): print(f"{kn} grid {g} {c}: launches {len(v)} last {v[-1]:.6g}")
PY
}
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run b SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY
run c FETCH_SIZE
grep '"pq"' $O/a.out | cut -c1-300
