#!/bin/bash
# round 4, GPU call J: re-run of the fixed rows8 test, the C++ mirror incl. the product quantiser, randomised parity incl. PQ rounds
mkdir -p gpurun_out/r04j
O=gpurun_out/r04j
timeout 900 python -m pytest tests/test_gpu_rows8.py tests/test_gpu_cpp_mirror.py tests/test_gpu_pq.py -m gpu -q --timeout=600 > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/tests.txt; tail -6 $O/tests.txt
timeout 400 python tools/fuzz_parity.py 150 40401 > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
