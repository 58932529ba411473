#!/bin/bash
# tools/build_variant.sh <name> [-D...]: build ab/libmoonrt_<name>.so with extra defines (A/B experiments only)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/ab"
cd "$ROOT/moonrtx_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
  -fhip-fp32-correctly-rounded-divide-sqrt "$@" mrtx_kernels.hip mrtx_api.hip -o "$ROOT/ab/libmoonrt_$NAME.so"
echo "built ab/libmoonrt_$NAME.so"
