#!/bin/bash
# Sweep of the megakernel's wave-scheduling knobs (speed only; results are bit-identical at every setting).
#   CRUCIBLE_WALK_ROUND  wrappers a lane steps through before parked leaves are intersected (0 = unbounded)
#   CRUCIBLE_WALK_EXIT   leave the walk once this many of the 64 lanes are done walking
for k in ${ROUNDS:-0 8 12 16}; do for t in ${EXITS:-64 56 48}; do
  echo -n "round=$k exit=$t : "
  CRUCIBLE_WALK_ROUND=$k CRUCIBLE_WALK_EXIT=$t python scripts/gpu_bench_quick.py ${W:-1920} ${SPP:-64} ${MODES:-f32} 2>&1 | grep -E "^f(32|64)" | cut -c1-75 | tr '\n' '|'; echo
done; done
