#!/bin/bash
# threshold sweep for the wave scheduler (speed only)
for st in ${STS:-8 16 24 32 40 48}; do for lt in ${LTS:-8 16 32}; do
  echo -n "shade=$st leaf=$lt : "
  CRUCIBLE_SHADE_THRESHOLD=$st CRUCIBLE_LEAF_THRESHOLD=$lt python scripts/gpu_bench_quick.py ${W:-1920} ${SPP:-32} ${MODES:-f32} 2>&1 | grep -E "^f(32|64)" | cut -c1-75 | tr '\n' '|'; echo
done; done
