#!/bin/bash
# C3 (8x512 bf16, 100 000 samples) step / k16 time of two library builds, with and without the stash traffic (BRIEF_DIAG=1:
# zero-record descriptors, results wrong, instruction stream unchanged):  tools/ab_c3.sh libA.so libB.so
for rep in 1 2; do for lib in "$@"; do for diag in 0 1; do
  BRIEF_DIAG=$diag BRIEF_LIB=$PWD/brief_pytorch_amd/$lib python bench.py --config c3 --precision bf16 --no-extras --no-cpu-baseline --no-psnr --steps 100 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith(chr(123)):
        o=json.loads(l); print('$lib diag=$diag: ms_per_step %.4f k16_ms %.4f' % (o['ms_per_step'], o['roofline']['kernel_ms']))
"
done; done; done
