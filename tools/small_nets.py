"""scratch: small-net regime timing (C1 2x64 full batch 64^3, default.yaml 5x22, C5-like 7x56 randompoint)"""
import sys, torch
sys.path.insert(0, '.')
from brief_pytorch_amd.fit import Fitter
from brief_pytorch_amd.networks import SIREN
def run(L, F, dims, sampler, n):
    torch.manual_seed(0)
    pop = dims[0] * dims[1] * dims[2]
    m = SIREN(features=F, layers=L, w0=20).to('cuda')
    tv = torch.rand(pop, 1, device='cuda') * 100
    fit = Fitter(m, tv, dims, sampler=sampler, sample_size=n)
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.6:          # past the clock ramp after idle (tools/clock_ramp.py)
        fit.run(100)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fit.run(500)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 500
    M = 3 * F + (L - 2) * F * F + F
    print("L=%d F=%d n=%d: %.3f ms/step %.1f Msamples/s %.1f TFLOP/s" % (L, F, fit.n, ms, fit.n / ms / 1e3, 2 * (3 * M - 3 * F) * fit.n / ms / 1e9), flush=True)
big = len(sys.argv) > 1 and sys.argv[1] == 'all'
run(3, 64, (64, 64, 64), 'full', 0)
run(5, 22, (64, 64, 64), 'full', 0)
run(5, 35, (128, 128, 128), 'randompoint', 100000)
run(7, 56, (64, 256, 256), 'randompoint', 100000)
run(9, 30, (64, 256, 256), 'randompoint', 100000)
if big:
    run(9, 512, (128, 128, 128), 'randompoint', 100000)
    run(5, 256, (256, 256, 256), 'randompoint', 100000)
