"""scratch: k_fused time against the number of 32-sample tiles per persistent slot (tail quantisation)"""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from brief_pytorch_amd import _lib
from brief_pytorch_amd.fit import Fitter
from brief_pytorch_amd.networks import SIREN
for n in (98304, 100000, 114688, 81920):
    torch.manual_seed(0)
    m = SIREN(features=256, layers=5, w0=20).to('cuda')
    tv = torch.rand(256 ** 3, 1, device='cuda') * 100
    fit = Fitter(m, tv, (256, 256, 256), sample_size=n)
    for _ in range(5): fit.step()
    torch.cuda.synchronize()
    _lib.check(_lib.lib().brief_profile_enable(1))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): fit.step()
    e1.record(); torch.cuda.synchronize()
    t, k = C.c_double(0), C.c_int64(0)
    _lib.lib().brief_profile_fused(C.byref(t), C.byref(k))
    _lib.check(_lib.lib().brief_profile_enable(0))
    step = e0.elapsed_time(e1) / 40
    kf = t.value / k.value
    print("n=%d tiles=%d (%.2f per slot): step %.3f ms  k_fused %.3f ms  (%.1f TF, frac %.3f)  rest %.3f" % (
        n, (n + 31) // 32, (n + 31) // 32 / 512.0, step, kf, 791040.0 * n / kf / 1e9, 791040.0 * n / kf / 1e9 / 157.3, step - kf), flush=True)
