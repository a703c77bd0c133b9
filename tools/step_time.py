"""step time of one net shape for several batch sizes: python tools/step_time.py L F prec n1,n2,... [steps]
(per-batch-size: ms/step and ns/sample; used to see tile-round quantisation and cache-residency effects)"""
import sys, time, torch
sys.path.insert(0, '.')
from brief_pytorch_amd.fit import Fitter
from brief_pytorch_amd.networks import SIREN
L, F, prec = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
ns = [int(v) for v in sys.argv[4].split(',')]
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 100
tv = torch.rand(256 ** 3, 1, device='cuda') * 100
for n in ns:
    torch.manual_seed(0)
    m = SIREN(features=F, layers=L, w0=20, precision=prec).to('cuda')
    fit = Fitter(m, tv, (256, 256, 256), sampler='randompoint', sample_size=n)
    fit.run(int(__import__("os").environ.get("STEP_WARM", "600")))          # past the clock ramp after idle (tools/clock_ramp.py)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fit.run(steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("%dx%d %s n=%d: %.4f ms/step, %.3f ns/sample" % (L - 1, F, prec, n, dt * 1e3, dt * 1e9 / n), flush=True)
