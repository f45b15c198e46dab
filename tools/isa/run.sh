#!/bin/bash
# tools/isa/run.sh [WALKS_SET bitmask] [extra -D flags ...] -> /tmp/walks.s + the per-kernel static profile (tools/isa_loops.py)
set -e
cd "$(dirname "$0")/../.."
SET=${1:-15}; shift || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero --cuda-device-only -S \
  -Rpass-analysis=kernel-resource-usage -DWALKS_SET=$SET "$@" -I coltt_amd/csrc -I include tools/isa/walks.hip -o /tmp/walks.s 2> /tmp/walks.rpass
python tools/isa_loops.py /tmp/walks.s
python tools/kernel_resources.py /tmp/walks.rpass
