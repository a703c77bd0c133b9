sed -i 's/fit.run(30)/fit.run(600)/' tools/step_time.py
python3 tools/step_time.py 5 256 fp32 98304,100000,114688,100000 1000
BRIEF_DIAG=1 python3 tools/step_time.py 5 256 fp32 98304,100000 1000
BRIEF_LIB=$PWD/brief_pytorch_amd/libbrief_hip_stamps.so STAMP_STEPS=600 python3 tools/read_stamps.py
