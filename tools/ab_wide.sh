#!/bin/bash
# interleaved A/B of library builds on the widths above 1024 features (k_wide): tools/ab_wide.sh "5 1495,2048 100000 8 12" libA.so libB.so ...
args="$1"; shift
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib (round $rep)"
    BRIEF_LIB=$PWD/$lib python tools/width_sweep.py $args 2>&1 | grep -v amdgpu.ids
  done
done
