#!/bin/bash
# tools/one_kernel.sh [-D...]: compile ONE kernel instantiation (default render_kernel<64, false, true, 2, false>; -DMRTX_DEV_ONE_MODE=0
# for the direct kernel, =9 for path_kernel<false, true>) to /tmp/one.s and print its resource usage -- seconds, no GPU needed
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/moonrtx_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
  -fhip-fp32-correctly-rounded-divide-sqrt -S --cuda-device-only -DMRTX_DEV_ONE "$@" \
  -Rpass-analysis=kernel-resource-usage mrtx_kernels.hip -o ${OUT:-/tmp/one.s} 2> /tmp/one.log || { tail -30 /tmp/one.log; exit 1; }
python "$ROOT/tools/resource_usage.py" /tmp/one.log
