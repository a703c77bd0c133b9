"""scratch: timing ablations of the train step (BRIEF_DEBUG flags); results are invalid when flags are set"""
import os, sys, subprocess
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, '.')
    from brief_pytorch_amd import _lib
    from brief_pytorch_amd.networks import SIREN
    from brief_pytorch_amd.fit import Fitter
    torch.manual_seed(0)
    pop = 256**3
    m = SIREN(features=256, layers=5, w0=20).to('cuda')
    tv = torch.rand(pop, 1, device='cuda') * 100
    fit = Fitter(m, tv, (256,256,256), sample_size=int(sys.argv[2]) if len(sys.argv) > 2 else 100000)
    for _ in range(5): fit.step()
    torch.cuda.synchronize()
    _lib.check(_lib.lib().brief_profile_enable(1))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fit.step()
    e1.record(); torch.cuda.synchronize()
    import ctypes as C
    t, n = C.c_double(0), C.c_int64(0)
    _lib.lib().brief_profile_fused(C.byref(t), C.byref(n))
    print("dbg=%s n=%d: step %.3f ms, k_fused %.3f ms" % (os.environ.get('BRIEF_DEBUG', '0'), fit.n, e0.elapsed_time(e1) / 30, t.value / n.value), flush=True)
else:
    import shutil
    for stg in ('0', '1', '2', '3', '5'):
        print('wpe2 stagger', stg, flush=True)
        subprocess.call([sys.executable, __file__, 'run'], env={**os.environ, 'BRIEF_STAGGER': stg})
    subprocess.call([sys.executable, __file__, 'run'], env={**os.environ, 'BRIEF_STAGGER': '2', 'BRIEF_DEBUG': '7'})
    shutil.copy('brief_pytorch_amd/libbrief_hip_wpe3.so', 'brief_pytorch_amd/libbrief_hip.so')
    for stg in ('0', '1', '2'):
        print('wpe3 stagger', stg, flush=True)
        subprocess.call([sys.executable, __file__, 'run'], env={**os.environ, 'BRIEF_STAGGER': stg})
    print('wpe3 build with 2 wg/cu', flush=True)
    subprocess.call([sys.executable, __file__, 'run'], env={**os.environ, 'BRIEF_STAGGER': '2', 'BRIEF_WG_PER_CU': '2'})
