#!/bin/bash
# headline (4x256 fp32) step / k_fused time of several library builds, interleaved:  tools/ab_c2.sh libA.so libB.so ...
for rep in 1 2 3; do for lib in "$@"; do
  BRIEF_LIB=$PWD/brief_pytorch_amd/$lib python bench.py --no-extras --no-cpu-baseline --no-psnr --steps 300 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith(chr(123)):
        o=json.loads(l); print('$lib: ms_per_step %.4f k_fused_ms %.4f value %.2fM' % (o['ms_per_step'], o['roofline']['kernel_ms'], o['value']/1e6))
"
done; done
