"""N train steps WITHOUT the fused optimizer (brief_siren_train_step: k_fused + k_wgrad + k_reduce with update = 0), for rocprofv3: what of
k_reduce's time is the slab sum and what the optimizer + packed write-through:  python tools/one_trainstep.py [steps] [precision]"""
import sys, torch
sys.path.insert(0, '.')
from brief_pytorch_amd.networks import SIREN
torch.manual_seed(0)
m = SIREN(features=256, layers=5, w0=20, precision=sys.argv[2] if len(sys.argv) > 2 else 'fp32').to('cuda')
tv = torch.rand(256 ** 3, 1, device='cuda') * 100
idx = torch.randint(0, 256 ** 3, (100000,), device='cuda')
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
    m.train_step(100000, tv, idx=idx, grid=((256, 256, 256), -1.0, 1.0))
torch.cuda.synchronize()
