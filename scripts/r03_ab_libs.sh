#!/bin/bash
# A/B of library builds under scripts/diag/ (git-ignored) against the in-tree one: LIBS="a.so b.so" WORK="book1:f64:1920:512 ..."
set -o pipefail
mkdir -p gpurun_out
OUT=gpurun_out/${TAG:-r03_ab_libs}.txt
: > $OUT
for w in ${WORK:-book1:f64:1920:512 million:f64:3840:64 teapot:f64:1920:256}; do
  IFS=: read wl real width spp <<< "$w"
  for lib in "" $LIBS; do
    if [ -z "$lib" ]; then python scripts/ab_render.py $wl $real $width $spp 2>/dev/null | tee -a $OUT
    else LIB=$PWD/scripts/diag/$lib python scripts/ab_render.py $wl $real $width $spp 2>/dev/null | tee -a $OUT; fi
  done
done
