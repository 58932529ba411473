#!/bin/bash
# Occupancy sensitivity of the render kernel: dynamic LDS per workgroup (MOONRT_DEV_LDS) caps the one-wave workgroups a CU holds
# (160 KB of LDS: 13312 B -> 12 = 3 per SIMD, 10000 B -> 16 = 4 per SIMD).  usage: [LIB=ab/libmoonrt_x.so] [BENCH_ARGS=...] tools/occ_sweep.sh <bytes>...
cd $GRAFT_REPO_ROOT
[ -n "$LIB" ] && export MOONRT_LIB=$GRAFT_REPO_ROOT/$LIB
for lds in "$@"; do
  MOONRT_DEV_LDS=$lds python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary $BENCH_ARGS 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lds $lds', 'frame', d['ms_per_step'], 'render', d['primary_ms'], 'paths', d['paths_ms'])"
done
