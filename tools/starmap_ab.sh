#!/bin/bash
# A/B builds on the star-map frame (the reference's default environment): tools/starmap_ab.sh lib1.so lib2.so ... (paths relative to the repo root)
cd $GRAFT_REPO_ROOT
for round in $(seq 1 ${ROUNDS:-2}); do for lib in "$@"; do
  MOONRT_LIB=$GRAFT_REPO_ROOT/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['also_starmap']; print('$lib round $round: headline', d['ms_per_step'], ' star map', s['ms_per_step'], 'ms (render', s['primary_ms'], 'paths', s['paths_ms'], ')')"
done; done
