#!/bin/bash
# k_reduce threads per hidden parameter behind k_small (BRIEF_REDUCE_SG), one box, two repetitions
for rep in 1 2; do for sg in 8 16 32 64; do echo "== BRIEF_REDUCE_SG=$sg"; BRIEF_REDUCE_SG=$sg python tools/small_nets.py 2>&1 | grep "L="; done; done
