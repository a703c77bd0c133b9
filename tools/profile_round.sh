#!/bin/bash
# Collect the judged profile of the bench workload on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r01
# kernel trace + stats, three separate PMC passes (never combined with other trace domains), the bench lines.
set -o pipefail
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o p -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-psnr --no-extras > $out/bench_trace_run.json 2> $out/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o p -- python3 bench.py --steps 10 --warmup 2 --preroll 20 --preroll-seconds 0 --no-cpu-baseline --no-psnr --no-extras > /dev/null 2> $out/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o p -- python3 bench.py --steps 10 --warmup 2 --preroll 20 --preroll-seconds 0 --no-cpu-baseline --no-psnr --no-extras > /dev/null 2> $out/write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/sq -o p -- python3 bench.py --steps 10 --warmup 2 --preroll 20 --preroll-seconds 0 --no-cpu-baseline --no-psnr --no-extras > /dev/null 2> $out/sq.err
# the line the DRIVER records: its flags, everything else default (configs, sweeps, wall clocks, cpu_baseline)
python3 bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err
# ... and the 200-step headline alone
python3 bench.py --no-extras --no-cpu-baseline --no-psnr > $out/bench_200.json 2> $out/bench_200.err
echo done
