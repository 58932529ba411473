#!/bin/bash
# rocprofv3 evidence for the path_seg_range (2,4) frame (render_kernel<MODE 2> + path_kernel + resolve_paths_kernel).
# usage: tools/profile_paths.sh <tag> [bench args]; kernel-trace/stats and every PMC group are SEPARATE passes.
set -o pipefail
TAG=${1:-paths}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --path-seg 2 4 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || exit 1
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- $BENCH > $OUT/g$i.log 2>&1 || echo "group $i ($grp) failed"
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
def short(n):
    for k in ("path_kernel", "resolve_paths_kernel", "render_kernel"):
        if k in n: return k + ("<stats>" if ("<64, true" in n or "path_kernel<true" in n) else "")
    return None
for fn in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = short(r["Kernel_Name"])
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print("==", k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"  {c:45s} {sum(v)/len(v):16.5e}  n={len(v)}")
for fn in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats"); print(open(fn).read())
PY
cat $OUT/summary.txt
