#!/bin/bash
# A/B of the overlapped path stage (MOONRT_PATH_OVERLAP sub-parts, MOONRT_PATH_OVERLAP_WAVES persistent waves while sharing the chip)
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary $BENCH_ARGS 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('overlap $1 waves $2:', d['value'], 'Mrays/s  frame', d['ms_per_step'], 'ms  render', d['primary_ms'], ' paths(exposed)', d['paths_ms'])"; }
MOONRT_PATH_OVERLAP=0 run 0 -
for ov in ${OVS:-4 8 16}; do for w in ${WAVES:-1024 2048 3072}; do MOONRT_PATH_OVERLAP=$ov MOONRT_PATH_OVERLAP_WAVES=$w run $ov $w; done; done
MOONRT_PATH_OVERLAP=0 run 0 -
