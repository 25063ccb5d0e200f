#!/bin/bash
# Re-sweep of the walk's scheduling knobs after the box step got cheaper (round 3): rounds of R steps, leave at E done lanes,
# leaf phase waits for L parked lanes.  WORK="book1:f64:1920:128" etc.  -> gpurun_out/r03_knob_sweep.txt
set -o pipefail
mkdir -p gpurun_out
OUT=gpurun_out/${TAG:-r03_knob_sweep}.txt
: > $OUT
for w in ${WORK:-book1:f64:1920:128}; do
  IFS=: read wl real width spp <<< "$w"
  for knobs in ${KNOBS:-10:56:8 12:56:8 14:56:8 16:56:8 10:60:8 12:60:8 10:48:8 10:56:12 10:56:16 12:56:12 14:60:12 8:56:8}; do
    IFS=: read R E L <<< "$knobs"
    echo -n "R=$R E=$E L=$L " | tee -a $OUT
    CRUCIBLE_WALK_ROUND=$R CRUCIBLE_WALK_EXIT=$E CRUCIBLE_WALK_LEAF_MIN=$L python scripts/ab_render.py $wl $real $width $spp 2>/dev/null | tee -a $OUT
  done
done
