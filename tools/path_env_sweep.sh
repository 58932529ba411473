#!/bin/bash
# path-stage scheduling knobs on the whole cfg3 frame (one GPU): tools/path_env_sweep.sh
cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', 'frame', d['ms_per_step'], 'render', d['primary_ms'], 'paths', d['paths_ms'])"; }
run X=1
run MOONRT_PATH_GRP=2
run MOONRT_PATH_GRP=4
run MOONRT_PATH_NSUB=2
run MOONRT_PATH_NSUB=8
run MOONRT_PATH_REFILL=24
run MOONRT_PATH_REFILL=40
run MOONRT_PATH_SEGMIN=8
run MOONRT_PATH_SEGMIN=24
run MOONRT_PATH_HITMIN=8
run MOONRT_PATH_HITMIN=24
