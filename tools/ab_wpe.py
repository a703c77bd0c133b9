"""scratch: does a third resident k_fused workgroup per CU help where the registers allow it without spills?  5x128 (NT = 4: 139 VGPRs) and 5x96 (NT = 3)
   BRIEF_LIB=... BRIEF_WG_PER_CU=2|3 python tools/ab_wpe.py"""
import sys, time, torch
sys.path.insert(0, '.')
from brief_pytorch_amd.fit import Fitter
from brief_pytorch_amd.networks import SIREN
for L, F in ((5, 128), (5, 96), (5, 160), (5, 192), (5, 224), (5, 256)):
    torch.manual_seed(0)
    m = SIREN(features=F, layers=L, w0=20).to('cuda')
    tv = torch.rand(256 ** 3, 1, device='cuda') * 100
    fit = Fitter(m, tv, (256, 256, 256), sampler='randompoint', sample_size=100000)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.6:
        fit.run(100); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fit.run(400); e1.record(); torch.cuda.synchronize()
    print("L=%d F=%d: %.4f ms/step" % (L, F, e0.elapsed_time(e1) / 400), flush=True)
