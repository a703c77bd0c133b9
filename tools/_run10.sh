bash tools/profile_round.sh r02 > gpurun_out/prof_r02.log 2>&1
bash tools/profile_bf16.sh r02 c3 > gpurun_out/prof16_r02.log 2>&1
bash tools/profile_small.sh r02 > gpurun_out/prof_small_r02.log 2>&1
./build/coexec > gpurun_out/r02_coexec.txt 2>&1
./build/clock > gpurun_out/r02_clock.txt 2>&1
python3 tools/read_stamps.py > gpurun_out/r02_stamps_fused.txt 2>&1
python3 tools/read_stamps16.py 9 512 > gpurun_out/r02_stamps_k16.txt 2>&1
python3 tools/read_stamps_small.py 5 22 > gpurun_out/r02_stamps_small.txt 2>&1
tail -2 gpurun_out/prof_r02.log gpurun_out/prof16_r02.log gpurun_out/prof_small_r02.log
