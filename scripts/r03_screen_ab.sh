#!/bin/bash
# Round 3: f32-screened slab test -- parity subset, then A/B (CRUCIBLE_SCREEN=0|1) on the frames whose tree is read from global memory.
set -o pipefail
mkdir -p gpurun_out
CRUCIBLE_SCREEN=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_relaxed.py tests/test_gpu_refit.py tests/test_gpu_lists.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r03_screen_tests.txt 2>&1 || { tail -40 gpurun_out/r03_screen_tests.txt; exit 1; }
tail -3 gpurun_out/r03_screen_tests.txt
{
for scr in 0 1; do
  echo "== CRUCIBLE_SCREEN=$scr"
  for w in "book1 f64 1920 512" "teapot f64 1920 256" "million f64 3840 64" "movie f64 1920 128"; do
    CRUCIBLE_SCREEN=$scr python scripts/ab_render.py $w 2>/dev/null
  done
done
} 2>&1 | tee gpurun_out/r03_screen_ab.txt
