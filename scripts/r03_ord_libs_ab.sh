#!/bin/bash
# A/B of library builds (scripts/diag/*.so against the in-tree one) on the opt-in ordered tree (BVH=ordered): LIBS="a.so" WORK="book1:f64:1920:512 ..."
set -o pipefail
mkdir -p gpurun_out
OUT=gpurun_out/${TAG:-r03_ord_libs_ab}.txt
: > $OUT
for w in ${WORK:-book1:f64:1920:512 teapot:f64:1920:256 million:f64:3840:64}; do
  IFS=: read wl real width spp <<< "$w"
  for lib in "" $LIBS; do
    if [ -z "$lib" ]; then BVH=ordered python scripts/ab_render.py $wl $real $width $spp 2>/dev/null | tee -a $OUT
    else BVH=ordered LIB=$PWD/scripts/diag/$lib python scripts/ab_render.py $wl $real $width $spp 2>/dev/null | tee -a $OUT; fi
  done
done
