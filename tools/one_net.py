"""scratch: N steps of one net shape (for rocprofv3 counter passes): python tools/one_net.py L F d h w sampler n steps"""
import sys, torch
sys.path.insert(0, '.')
from brief_pytorch_amd.fit import Fitter
from brief_pytorch_amd.networks import SIREN
L, F, d, h, w = (int(v) for v in sys.argv[1:6])
sampler, n, steps = sys.argv[6], int(sys.argv[7]), int(sys.argv[8])
torch.manual_seed(0)
m = SIREN(features=F, layers=L, w0=20).to('cuda')
tv = torch.rand(d * h * w, 1, device='cuda') * 100
fit = Fitter(m, tv, (d, h, w), sampler=sampler, sample_size=n)
for _ in range(steps): fit.step()
torch.cuda.synchronize()
print("done", float(fit.step()))
