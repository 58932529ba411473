#!/bin/bash
# A/B several builds of libmoonrt.so on one GPU box: [BENCH_ARGS="..."] [ROUNDS=2] tools/ab_bench.sh lib1.so lib2.so ...
# (paths relative to the repo root, "default" = moonrtx_amd/libmoonrt.so); prints frame / render / path-stage milliseconds
cd $GRAFT_REPO_ROOT
for round in $(seq 1 ${ROUNDS:-2}); do
for lib in "$@"; do
  if [ "$lib" = "default" ]; then unset MOONRT_LIB; else export MOONRT_LIB=$GRAFT_REPO_ROOT/$lib; fi
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary $BENCH_ARGS 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'round $round', d['value'], 'Mrays/s  frame', d['ms_per_step'], 'ms  render', d['primary_ms'], ' paths', d['paths_ms'])"
done; done
