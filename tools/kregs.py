"""register / scratch / LDS footprint of the kernels in a hipcc -save-temps assembly listing:
   hipcc --offload-arch=gfx950 -O3 ... -save-temps=obj -o /tmp/x.so  &&  python tools/kregs.py /tmp/brief_hip-hip-amdgcn-amd-amdhsa-gfx950.s [pattern]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if pat and pat not in dem:
        continue
    g = lambda k: (re.search(r"\.amdhsa_%s (\S+)" % k, body) or [None, "?"])[1]
    print("%-60s vgpr %s accum_off %s sgpr %s scratch %s" % (dem[:60], g("next_free_vgpr"), g("accum_offset"), g("next_free_sgpr"), g("private_segment_fixed_size")))
