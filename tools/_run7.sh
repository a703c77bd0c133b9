python -m pytest tests/test_gpu_bf16.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "not 3000 and not C2 and not fp32" > gpurun_out/r02_t7.log 2>&1; echo rc=$? >> gpurun_out/r02_t7.log
tail -4 gpurun_out/r02_t7.log
BRIEF_K16_NW=8 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q -k "track_fp32" > gpurun_out/r02_t7b.log 2>&1; echo rc=$? >> gpurun_out/r02_t7b.log
tail -2 gpurun_out/r02_t7b.log
export TMPDIR=/tmp
for nw in 4 8; do
  export BRIEF_K16_NW=$nw
  python tools/step_time.py 9 512 bf16 100000 30 2>&1 | grep -v amdgpu
  python tools/step_time.py 5 256 bf16 100000 60 2>&1 | grep -v amdgpu
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/v2_${nw} -o p -- python3 tools/one_net16.py 9 512 bf16 20 > /dev/null 2>&1
  python3 - <<PY
import csv
for r in list(csv.DictReader(open('gpurun_out/v2_${nw}/p_kernel_stats.csv')))[:5]:
    print('nw=$nw  %-42s calls %s avg %.1f us' % (r['Name'][:42], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
python tools/step_time.py 5 22 fp32 262144 200 2>&1 | grep -v amdgpu
python tools/step_time.py 3 64 fp32 262144 200 2>&1 | grep -v amdgpu
python tools/step_time.py 7 56 fp32 100000 200 2>&1 | grep -v amdgpu
