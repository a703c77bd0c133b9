#!/bin/bash
# interleaved A/B of library builds on step time + fused-kernel time of given nets: tools/ab_step.sh "5 256 100000 300" libA.so libB.so ...   (two rounds)
args="$1"; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    echo "== $lib (round $rep)"
    BRIEF_LIB=$PWD/$lib python tools/width_sweep.py $args 2>&1 | grep -v amdgpu.ids
  done
done
