#!/bin/bash
# Kernel trace + HBM-traffic counters of the bf16 path (BASELINE config 3: 8x512, and 4x256) on the GPU box:
#   bash tools/profile_bf16.sh r02 c3 [bf16|bf16x3]     (run through gpurun from the repo root)
# separate --pmc passes, never combined with other trace domains.
set -o pipefail
tag=${1:-r02}
cfg=${2:-c3}
prec=${3:-bf16}
out=gpurun_out/prof16_${tag}_${cfg}_${prec}
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o p -- python3 bench.py --config $cfg --precision $prec --steps 40 --warmup 5 --no-cpu-baseline --no-psnr --no-extras > $out/bench_trace_run.json 2> $out/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o p -- python3 bench.py --config $cfg --precision $prec --steps 6 --warmup 2 --preroll 2 --preroll-seconds 0 --no-cpu-baseline --no-psnr --no-extras > /dev/null 2> $out/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o p -- python3 bench.py --config $cfg --precision $prec --steps 6 --warmup 2 --preroll 2 --preroll-seconds 0 --no-cpu-baseline --no-psnr --no-extras > /dev/null 2> $out/write.err
python3 tools/pmc_summary.py $out/fetch/p_counter_collection.csv $out/write/p_counter_collection.csv > $out/pmc.txt 2>&1
echo done
