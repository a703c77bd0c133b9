"""scratch: N steps of one net shape, chosen precision: python tools/one_net16.py L F prec steps"""
import sys, torch
sys.path.insert(0, '.')
from brief_pytorch_amd.fit import Fitter
from brief_pytorch_amd.networks import SIREN
L, F, prec, steps = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
torch.manual_seed(0)
m = SIREN(features=F, layers=L, w0=20, precision=prec).to('cuda')
tv = torch.rand(256 ** 3, 1, device='cuda') * 100
fit = Fitter(m, tv, (256, 256, 256), sampler='randompoint', sample_size=100000)
for _ in range(steps): fit.step()
torch.cuda.synchronize()
print("done", float(fit.step()))
