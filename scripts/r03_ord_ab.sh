#!/bin/bash
# ordered-tree screening records: BVH-mode tests + A/B with CRUCIBLE_SCREEN=0|1 on the opt-in trees
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bvh_modes.py tests/test_gpu_relaxed.py tests/test_gpu_parity.py tests/test_gpu_refit.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r03_ord_tests.txt 2>&1 || { tail -30 gpurun_out/r03_ord_tests.txt; exit 1; }
tail -1 gpurun_out/r03_ord_tests.txt
for scr in 0 1; do
  echo "== CRUCIBLE_SCREEN=$scr"
  for w in "book1 f64 1920 512" "teapot f64 1920 256" "million f64 3840 64"; do
    for b in ordered sah; do BVH=$b CRUCIBLE_SCREEN=$scr python scripts/ab_render.py $w 2>/dev/null; done
  done
  CRUCIBLE_SCREEN=$scr python scripts/ab_render.py book1 f64 1920 512 2>/dev/null
done | tee gpurun_out/r03_ord_ab.txt
