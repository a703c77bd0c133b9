for lib in libbrief_hip.so libbrief_hip_apre1.so libbrief_hip_apre2.so; do
  export BRIEF_LIB=$PWD/brief_pytorch_amd/$lib
  echo "== $lib"
  python tools/step_time.py 9 512 bf16 100000 30 2>&1 | grep -v amdgpu
  python tools/step_time.py 5 256 bf16 100000 60 2>&1 | grep -v amdgpu
done
