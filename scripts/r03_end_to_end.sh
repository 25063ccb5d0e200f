#!/bin/bash
# Round 3, VERDICT item 4(a): the reference's own benchmark shape -- book1 400x225 @ 100 spp, depth 50, the whole of
# render_scene (handle, BVH build, render, P3 file) -- through the compiled host, cold (run 0) and warm (runs 1..4).
set -o pipefail
mkdir -p gpurun_out /tmp/r03e2e
OUT=gpurun_out/r03_end_to_end_criterion.txt
: > $OUT
for real in f64 f32; do
  for fmt in ppm p6 png; do
    echo "# crucible_render --world 1 --width 400 --samples 100 --real $real --format $fmt --repeat 5 --timing" >> $OUT
    ./crucible_amd/host/crucible_render --file /tmp/r03e2e/crit_${real}_${fmt} --world 1 --width 400 --samples 100 --real $real --format $fmt --repeat 5 --timing 2>/dev/null >> $OUT || exit 1
  done
done
echo "# 1920x1080 @ 512 (configs[1]) end to end, f64, ppm" >> $OUT
./crucible_amd/host/crucible_render --file /tmp/r03e2e/c2 --world 1 --width 1920 --samples 512 --real f64 --format ppm --repeat 3 --timing 2>/dev/null >> $OUT || exit 1
cat $OUT
