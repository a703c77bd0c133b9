"""N steps of the 4x256 net in BRIEF_PREC_BF16X3 (for rocprofv3 passes): python tools/one_x3.py [steps]"""
import sys, torch
sys.path.insert(0, '.')
from brief_pytorch_amd.fit import Fitter
from brief_pytorch_amd.networks import SIREN
torch.manual_seed(0)
m = SIREN(features=256, layers=5, w0=20, precision='bf16x3').to('cuda')
tv = torch.rand(256 ** 3, 1, device='cuda') * 100
fit = Fitter(m, tv, (256, 256, 256), sampler='randompoint', sample_size=100000)
fit.run(int(sys.argv[1]) if len(sys.argv) > 1 else 300)
torch.cuda.synchronize()
