#!/bin/bash
# two campaign lanes side by side (two processes on the GPU): tools/fuzz_campaign2.sh <first seed> <seeds per lane> [cases per seed]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( tools/fuzz_campaign.sh $1 $2 ${3:-300} > gpurun_out/fz_lane_a.log 2>&1 ) &
( sleep 5; tools/fuzz_campaign.sh $(($1 + 50)) $2 ${3:-300} > gpurun_out/fz_lane_b.log 2>&1 ) &
wait
cat gpurun_out/fz_lane_a.log gpurun_out/fz_lane_b.log
