#!/bin/bash
# Vector-memory path counters of bench.py's render kernel (TA / TCP / TD / UTCL1), one small --pmc pass each.
# usage: tools/profile_mem.sh <tag> [bench args]
set -o pipefail
TAG=${1:-mem}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary $*"
i=0
for grp in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TA_BUSY_avr TA_BUFFER_READ_WAVEFRONTS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_GATE_EN1_sum TCP_TOTAL_ACCESSES_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum" "TD_TC_STALL_sum TD_SPI_STALL_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- $BENCH > $OUT/g$i.log 2>&1 || echo "group $i ($grp) failed"
done
python3 - <<PY
import csv, glob, collections, os
acc = collections.defaultdict(list)
for fn in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "render_kernel<64, false" in r["Kernel_Name"] and (os.environ.get("KMATCH", "") in r["Kernel_Name"]):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(f"{k:45s} {sum(acc[k])/len(acc[k]):16.4e}  n={len(acc[k])}")
PY
