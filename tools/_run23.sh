mkdir -p gpurun_out/st
python3 tools/read_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/st/fp32.txt
python3 tools/read_stamps16.py 9 512 2>&1 | grep -v amdgpu.ids > gpurun_out/st/bf16.txt
python3 tools/read_stamps_small.py 5 22 2>&1 | grep -v amdgpu.ids > gpurun_out/st/small.txt
for d in 0 1; do BRIEF_DIAG=$d python3 tools/step_time.py 9 512 bf16 100000 400 2>&1 | grep -v amdgpu.ids | sed "s/^/diag=$d /"; done > gpurun_out/st/bf16_drop.txt
for d in 0 1; do BRIEF_DIAG=$d python3 tools/step_time.py 5 256 fp32 100000 1000 2>&1 | grep -v amdgpu.ids | sed "s/^/diag=$d /"; done > gpurun_out/st/fp32_drop.txt
cat gpurun_out/st/*.txt
