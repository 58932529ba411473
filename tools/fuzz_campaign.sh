#!/bin/bash
# Randomised bit-exact parity campaign on the GPU box: tools/fuzz_campaign.sh <first seed> <n seeds> [cases per seed]
# (appends one line per seed to gpurun_out/fuzz_campaign.log; stops at the first failing seed)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export MOONRT_DEFAULT_FLAGS=1
export MOONRT_PATH_QUEUE_MIN=${MOONRT_PATH_QUEUE_MIN:-0}   # small frames through the path queue as well (the library would keep them in the wave)
for ((s=$1; s<$1+$2; s++)); do
  timeout -k 10 300 python tools/fuzz_parity.py ${3:-300} $s > /tmp/fz_$$.log 2>&1 || { tail -3 /tmp/fz_$$.log | tee -a gpurun_out/fuzz_campaign.log; exit 1; }
  tail -1 /tmp/fz_$$.log | tee -a gpurun_out/fuzz_campaign.log
done
