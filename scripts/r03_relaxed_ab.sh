#!/bin/bash
# Round 3: relaxed sums (CR_SUM_RELAXED) -- parity tests, then A/B against the reference order on the BASELINE frames.
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_relaxed.py tests/test_gpu_parity.py -q -m gpu -k "relaxed or config0 or golden" > gpurun_out/r03_relaxed_tests.txt 2>&1 || { tail -40 gpurun_out/r03_relaxed_tests.txt; exit 1; }
tail -3 gpurun_out/r03_relaxed_tests.txt
{
for mode in reference relaxed; do
  echo "== CRUCIBLE_SUM_ORDER=$mode"
  CRUCIBLE_SUM_ORDER=$mode python scripts/ab_render.py book1 f64 1920 512
  CRUCIBLE_SUM_ORDER=$mode python scripts/ab_render.py book1 f32 1920 512
  CRUCIBLE_SUM_ORDER=$mode python scripts/ab_render.py teapot f64 1920 256
  CRUCIBLE_SUM_ORDER=$mode python scripts/ab_render.py million f64 3840 64
  CRUCIBLE_SUM_ORDER=$mode python scripts/ab_render.py movie f64 1920 128
  CRUCIBLE_SUM_ORDER=$mode python scripts/ab_render.py book1 f64 1920 64
done
} 2>&1 | tee gpurun_out/r03_relaxed_ab.txt
