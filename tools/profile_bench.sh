#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box.  Usage: tools/profile_bench.sh <tag> [extra bench args]
# kernel-trace/stats and each PMC group run as SEPARATE passes (never combined).  Then, on the build machine:
#   python tools/summarise_profile.py gpurun_out/prof_<tag> <tag>    (writes profiles/<tag>_*, profiles/traffic_latest.json)
set -o pipefail
TAG=${1:-r02}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary $*"
BENCH="python3 $GRAFT_REPO_ROOT/bench.py $ARGS"
python3 - <<PY > $OUT/meta.json
import json, sys
args = "$ARGS".split()
seg = [2, 4]
if "--path-seg" in args:
    i = args.index("--path-seg"); seg = [int(args[i + 1]), int(args[i + 2])]
json.dump({"command": "python3 bench.py $ARGS", "path_seg": seg}, sys.stdout)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- $BENCH > $OUT/pmc_l2.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1 || echo "pmc_sq2 failed (non-fatal)"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- python3 $GRAFT_REPO_ROOT/tools/calibrate_fetch.py > $OUT/cal_fetch.log 2>&1 || echo "calibration failed (non-fatal)"
find $OUT -name '*.csv' | head -50
