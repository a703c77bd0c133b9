for rep in 1 2; do for w in 1 2; do echo "BRIEF_WGRAD_PER_CU=$w"; BRIEF_WGRAD_PER_CU=$w python3 tools/step_time.py 5 256 fp32 100000 1000 2>&1 | grep -v amdgpu; done; done
