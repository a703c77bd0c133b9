set -o pipefail
python -m pytest tests/test_gpu_bf16.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not 3000 and not C2 and not fp32" > gpurun_out/r02_t4.log 2>&1; echo rc=$? >> gpurun_out/r02_t4.log
tail -4 gpurun_out/r02_t4.log
for nw in 4 8; do
  echo "NW=$nw"
  BRIEF_K16_NW=$nw python tools/step_time.py 9 512 bf16 100000 30
  BRIEF_K16_NW=$nw python tools/step_time.py 5 256 bf16 100000 60
done 2>&1 | grep -v amdgpu.ids
