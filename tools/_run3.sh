set -o pipefail
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not c3 and not 3000" > gpurun_out/r02_t3.log 2>&1; echo rc=$? >> gpurun_out/r02_t3.log
python bench.py --no-cpu-baseline --no-psnr > gpurun_out/r02_bench_b.json 2> gpurun_out/r02_bench_b.err
python tools/step_time.py 5 256 bf16 32768,98304,100000 60 > gpurun_out/r02_steptime.log 2>&1
python tools/step_time.py 9 512 bf16 32768,98304,100000 30 >> gpurun_out/r02_steptime.log 2>&1
python tools/step_time.py 5 256 fp32 32768,98304,100000 60 >> gpurun_out/r02_steptime.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/diag0 -o p -- python3 tools/one_net16.py 9 512 bf16 30 > /dev/null 2>&1
export BRIEF_DIAG=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/diag1 -o p -- python3 tools/one_net16.py 9 512 bf16 30 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/diag1f -o p -- python3 tools/one_net16.py 5 256 fp32 30 > /dev/null 2>&1
unset BRIEF_DIAG
tail -3 gpurun_out/r02_t3.log; cat gpurun_out/r02_steptime.log
for d in diag0 diag1 diag1f; do echo $d; head -6 gpurun_out/$d/p_kernel_stats.csv | cut -d, -f1-4 | cut -c1-100; done
