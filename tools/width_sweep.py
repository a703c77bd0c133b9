"""step / k_fused time and fraction of the fp32 MFMA peak over net widths: python tools/width_sweep.py L F1,F2,... [n] [steps] [pre-roll steps]
(algorithmic FLOPs of SURVEY 8d: train 2(3M - cin F) per sample; k_fused: forward 2M + dgrad + skinny gradients)"""
import ctypes as C
import sys
import torch
sys.path.insert(0, '.')
from brief_pytorch_amd import _lib
from brief_pytorch_amd.fit import Fitter
from brief_pytorch_amd.networks import SIREN

L = int(sys.argv[1])
Fs = [int(v) for v in sys.argv[2].split(',')]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 100
preroll = int(sys.argv[5]) if len(sys.argv) > 5 else 300
PEAK = 157.3e12
tv = torch.rand(256 ** 3, 1, device='cuda') * 100
for F in Fs:
    torch.manual_seed(0)
    m = SIREN(features=F, layers=L, w0=20).to('cuda')
    fit = Fitter(m, tv, (256, 256, 256), sampler='randompoint', sample_size=n)
    fit.run(preroll)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().brief_profile_enable(1))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fit.step()
    e1.record()
    torch.cuda.synchronize()
    t, k = C.c_double(0), C.c_int64(0)
    _lib.lib().brief_profile_fused(C.byref(t), C.byref(k))
    _lib.lib().brief_profile_enable(0)
    step = e0.elapsed_time(e1) / steps * 1e-3
    kf = t.value / max(k.value, 1) * 1e-3
    M = 3 * F + (L - 2) * F * F + F
    train = 2 * (3 * M - 3 * F)
    fused = 2 * M + 2 * (L - 2) * F * F + 2 * (3 * F + F)
    print("%dx%d n=%d: step %.4f ms (%.3f of peak, %.1f M samples/s)  k_fused %.4f ms (%.3f)" %
          (L - 1, F, n, step * 1e3, train * n / step / PEAK, n / step * 1e-6, kf * 1e3, fused * n / kf / PEAK), flush=True)
