"""scratch: cycle stamps of the diagnostic k16 build: python tools/read_stamps16.py L F"""
import os, sys
os.environ["BRIEF_LIB"] = os.path.abspath("brief_pytorch_amd/libbrief_hip_stamps.so")
import torch, numpy as np
sys.path.insert(0, '.')
from brief_pytorch_amd.networks import SIREN
from brief_pytorch_amd.fit import Fitter
L, F = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(0)
m = SIREN(features=F, layers=L, w0=20, precision='bf16').to('cuda')
tv = torch.rand(128 ** 3, 1, device='cuda') * 100
fit = Fitter(m, tv, (128, 128, 128), sample_size=100000)
for _ in range(int(os.environ.get("STAMP_STEPS", "400"))): fit.step()
torch.cuda.synchronize()
FP = 256 if F <= 256 else 512
npad = (100000 + 127) // 128 * 128
stash = (L - 1) * FP * npad // 2
rec_off = 2 * stash + 2 * (4 * npad // 2)      # phase + delta planes (round 3: no cosine plane)
st = m._ws[rec_off + 2 * 256 * 8: rec_off + 2 * 256 * 8 + 256 * 8 * 10].view(256 * 8, 10).cpu().numpy()
names = ['inputs+layer0', 'fwd chain', 'barrier after chain', 'fwd epilogue (sin/cos, stash, image)', 'barrier after image', 'head+loss+Wh^T g',
         'bwd epilogue (c load, mult, D stash)', 'bwd barriers + image', 'bwd chain', 'tile-end barrier']
tot = st.sum(1)
print('waves:', (tot > 0).sum(), 'mean total cycles/wave: %.0f' % tot[tot > 0].mean())
for i, nme in enumerate(names):
    v = st[tot > 0, i]
    print('%-40s mean %9.0f cycles  %5.1f%%   (min %9.0f max %9.0f)' % (nme, v.mean(), 100 * v.mean() / tot[tot > 0].mean(), v.min(), v.max()))
