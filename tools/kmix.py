"""static instruction mix of the kernels in a hipcc -save-temps assembly listing (whole function bodies: prologue + every loop once):
   python tools/kmix.py /tmp/brief_hip-hip-amdgcn-amd-amdhsa-gfx950.s [pattern]"""
import re, subprocess, sys
from collections import Counter
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for nm in re.findall(r'^(_Z\S+):', txt, re.M):
    dem = subprocess.run(['c++filt', nm], capture_output=True, text=True).stdout.strip()
    if pat not in dem or '.amdhsa_kernel ' + nm not in txt:
        continue
    start = txt.index('\n' + nm + ':')
    end = txt.index('.end_amdhsa_kernel', start)
    c = Counter()
    for l in txt[start:end].splitlines():
        t = l.strip().split()
        if not l.startswith('\t') or not t or t[0].startswith(('.', ';')):
            continue
        i = t[0]
        if i.startswith('v_mfma'): c['mfma'] += 1
        elif i in ('v_sin_f32', 'v_cos_f32') or i.startswith(('v_rcp', 'v_sqrt', 'v_exp', 'v_log', 'v_rsq')): c['trans'] += 1
        elif i.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')): c['lane'] += 1
        elif i.startswith('v_'): c['valu'] += 1
        elif i.startswith('ds_'): c['lds'] += 1
        elif i.startswith(('buffer_', 'global_', 'flat_', 'scratch_')): c['vmem'] += 1
        elif i.startswith('s_waitcnt'): c['wait'] += 1
        elif i.startswith('s_barrier'): c['barrier'] += 1
        elif i.startswith('s_'): c['salu'] += 1
        else: c['other'] += 1
    print("%-50s %s" % (dem[:50], dict(c)))
