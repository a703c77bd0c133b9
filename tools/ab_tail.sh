#!/bin/bash
# A/B of the k_fused launch plan (BRIEF_TAIL_ROUNDS single-tile rounds behind the persistent body), one box, interleaved
set -e
mkdir -p gpurun_out
for rep in 1 2; do
  for tr in 0 1 2 3; do
    BRIEF_TAIL_ROUNDS=$tr python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-psnr 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); print('tail_rounds=$tr rep=$rep ms_per_step=%.4f kernel_ms=%.4f frac=%.4f value=%.1fM' % (o['ms_per_step'], o['roofline']['kernel_ms'], o['roofline']['frac'], o['value']/1e6))
"
  done
done
