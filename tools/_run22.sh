export BRIEF_PACE=0
for m in 16 32 48 64; do echo "== static priority mode $((m/16))"; BRIEF_DIAG=$m python3 tools/wg_lifetimes.py 100000 2>&1 | grep -v amdgpu.ids | head -3; BRIEF_DIAG=$m python3 tools/wg_lifetimes.py 100000 2>&1 | grep "first slot"; done
for m in 0 16 32 48 64; do echo "== mode $((m/16))"; BRIEF_DIAG=$m python3 tools/step_time.py 5 256 fp32 98304,100000 1000 2>&1 | grep -v amdgpu.ids; done
