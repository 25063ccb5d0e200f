#!/bin/bash
# quick parity subset + the five BASELINE frames, five renders each (library as built; env selects variants)
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_relaxed.py -x -q -m gpu > gpurun_out/${TAG:-r03_quick}_tests.txt 2>&1 || { tail -30 gpurun_out/${TAG:-r03_quick}_tests.txt; exit 1; }
tail -1 gpurun_out/${TAG:-r03_quick}_tests.txt
for w in "book1 f64 1920 512" "book1 f32 1920 512" "teapot f64 1920 256" "million f64 3840 64" "movie f64 1920 128"; do
  python scripts/ab_render.py $w 2>/dev/null
done | tee gpurun_out/${TAG:-r03_quick}_ab.txt
CRUCIBLE_SUM_ORDER=reference python scripts/ab_render.py book1 f64 1920 512 2>/dev/null | tee -a gpurun_out/${TAG:-r03_quick}_ab.txt
