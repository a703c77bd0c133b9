"""scratch: interleaved A/B timing of library variants in ONE gpurun call (same device)"""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == 'run':
    import ctypes as C
    import torch
    sys.path.insert(0, '.')
    from brief_pytorch_amd import _lib
    from brief_pytorch_amd.fit import Fitter
    from brief_pytorch_amd.networks import SIREN
    torch.manual_seed(0)
    pop = 256 ** 3
    m = SIREN(features=256, layers=5, w0=20).to('cuda')
    tv = torch.rand(pop, 1, device='cuda') * 100
    fit = Fitter(m, tv, (256, 256, 256), sample_size=100000)
    for _ in range(5): fit.step()
    torch.cuda.synchronize()
    _lib.check(_lib.lib().brief_profile_enable(1))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): fit.step()
    e1.record(); torch.cuda.synchronize()
    t, n = C.c_double(0), C.c_int64(0)
    _lib.lib().brief_profile_fused(C.byref(t), C.byref(n))
    step = e0.elapsed_time(e1) / 40
    print("%-34s step %.3f ms  k_fused %.3f ms  rest %.3f ms" % (os.path.basename(os.environ.get('BRIEF_LIB', 'default')), step, t.value / n.value, step - t.value / n.value), flush=True)
else:
    libs = sys.argv[1:]
    for rnd in range(3):
        for lib in libs:
            subprocess.call([sys.executable, __file__, 'run'], env={**os.environ, 'BRIEF_LIB': os.path.abspath(lib)})
