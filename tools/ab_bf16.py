"""scratch: A/B of two library builds on the bf16 step and decode shapes"""
import os, subprocess, sys
for rnd in range(2):
    for lib in sys.argv[1:]:
        print("==", os.path.basename(lib), flush=True)
        env = {**os.environ, 'BRIEF_LIB': os.path.abspath(lib)}
        subprocess.call([sys.executable, "tools/bf16_timing.py", "bf16"], env=env)
        subprocess.call([sys.executable, "tools/decode_timing.py"], env=env)
