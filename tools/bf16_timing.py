"""scratch: step time of the bf16 path vs fp32 (C2 4x256 and C3 8x512 shapes, 100000 samples)"""
import sys, torch
sys.path.insert(0, '.')
from brief_pytorch_amd.fit import Fitter
from brief_pytorch_amd.networks import SIREN
def run(L, F, dims, n, prec, steps=30):
    torch.manual_seed(0)
    pop = dims[0] * dims[1] * dims[2]
    m = SIREN(features=F, layers=L, w0=20, precision=prec).to('cuda')
    tv = torch.rand(pop, 1, device='cuda') * 100
    fit = Fitter(m, tv, dims, sampler='randompoint', sample_size=n)
    for _ in range(5): fit.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fit.step()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    M = 3 * F + (L - 2) * F * F + F
    print("%s L=%d F=%d n=%d: %.3f ms/step %.1f Msamples/s %.1f TFLOP/s" % (prec, L, F, fit.n, ms, fit.n / ms / 1e3, 2 * (3 * M - 3 * F) * fit.n / ms / 1e9), flush=True)
precs = sys.argv[1:] or ['bf16', 'fp32']
for prec in precs:
    run(5, 256, (256, 256, 256), 100000, prec)
    run(9, 512, (256, 256, 256), 100000, prec)
