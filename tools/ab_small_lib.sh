#!/bin/bash
# narrow-net step times of several library builds, interleaved:  tools/ab_small_lib.sh libA.so libB.so
for rep in 1 2; do for lib in "$@"; do echo "== $lib"; BRIEF_LIB=$PWD/brief_pytorch_amd/$lib python tools/small_nets.py 2>&1 | grep "L="; done; done
