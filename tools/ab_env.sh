#!/bin/bash
# headline step time under values of one environment knob, interleaved:  tools/ab_env.sh NAME v1 v2 ...
name=$1; shift
for rep in 1 2; do for v in "$@"; do
  env $name=$v python bench.py --no-extras --no-cpu-baseline --no-psnr --steps 300 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith(chr(123)):
        o=json.loads(l); print('$name=$v: ms_per_step %.4f kernel_ms %.4f value %.2fM' % (o['ms_per_step'], o['roofline']['kernel_ms'], o['value']/1e6))
"
done; done
