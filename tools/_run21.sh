for p in 0 1; do echo "== BRIEF_PACE=$p"; BRIEF_PACE=$p python3 tools/wg_lifetimes.py 100000 2>&1 | grep -v amdgpu.ids | head -5; done
for p in 0 1; do echo "== BRIEF_PACE=$p"; BRIEF_PACE=$p python3 tools/step_time.py 5 256 fp32 98304,100000 1000 2>&1 | grep -v amdgpu.ids; done
