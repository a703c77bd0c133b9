#!/bin/bash
# kernel trace of the narrow-net shapes (k_small): bash tools/profile_small.sh r02   (through gpurun from the repo root)
set -o pipefail
tag=${1:-r02}
out=gpurun_out/prof_small_$tag
mkdir -p $out
export TMPDIR=/tmp
for cfg in "5 22 262144" "3 64 262144" "7 56 100000" "5 35 100000"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/L$1_F$2 -o p -- python3 tools/step_time.py $1 $2 fp32 $3 200 > $out/L$1_F$2.txt 2> $out/L$1_F$2.err
done
echo done
