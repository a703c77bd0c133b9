"""scratch: A/B timing of library variants on the small-net shapes, one process per (variant, round)"""
import os, subprocess, sys
libs = sys.argv[1:]
for rnd in range(2):
    for lib in libs:
        print("==", os.path.basename(lib), flush=True)
        subprocess.call([sys.executable, "tools/small_nets.py"], env={**os.environ, 'BRIEF_LIB': os.path.abspath(lib)})
