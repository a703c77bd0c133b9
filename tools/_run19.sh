export BRIEF_LIB=$PWD/brief_pytorch_amd/libbrief_hip_clock.so
export STAMP_STEPS=600
for d in 0 1 0 1; do echo "clock-only build, 600 steps, diag=$d"; BRIEF_DIAG=$d python3 tools/read_stamps.py 2>&1 | grep in-kernel; done
export BRIEF_LIB=$PWD/brief_pytorch_amd/libbrief_hip_stamps.so
echo "full stamps build, 600 steps"; python3 tools/read_stamps.py 2>&1 | grep in-kernel
