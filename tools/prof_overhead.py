"""what the live kernel timing (brief_profile_enable: an event pair bound to every k_fused dispatch) costs the step it measures:
step time of the headline fit with the timing off / on, interleaved.   python tools/prof_overhead.py [steps]"""
import ctypes as C
import sys
import torch
sys.path.insert(0, '.')
from brief_pytorch_amd import _lib
from brief_pytorch_amd.fit import Fitter
from brief_pytorch_amd.networks import SIREN

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tv = torch.rand(512 ** 3, 1, device='cuda') * 100
torch.manual_seed(0)
m = SIREN(features=256, layers=5, w0=20).to('cuda')
fit = Fitter(m, tv, (512, 512, 512), sampler='randompoint', sample_size=100000)
fit.run(400)
torch.cuda.synchronize()
L = _lib.lib()
_lib.check(L.brief_profile_enable(1)); _lib.check(L.brief_profile_enable(0))
for rep in range(3):
    for on in (0, 1):
        _lib.check(L.brief_profile_enable(on))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            fit.step()
        e1.record()
        torch.cuda.synchronize()
        t, k = C.c_double(0), C.c_int64(0)
        L.brief_profile_fused(C.byref(t), C.byref(k))
        _lib.check(L.brief_profile_enable(0))
        print("timing %s: %.4f ms per step%s" % ("on " if on else "off", e0.elapsed_time(e1) / steps, "  (k_fused %.4f ms over %d launches)" % (t.value / max(k.value, 1), k.value) if on else ""), flush=True)
