"""scratch: cycle stamps of the diagnostic k_small build (python tools/read_stamps_small.py L F)"""
import os, sys
os.environ["BRIEF_LIB"] = os.path.abspath("brief_pytorch_amd/libbrief_hip_stamps.so")
import torch, numpy as np
sys.path.insert(0, '.')
from brief_pytorch_amd.networks import SIREN
from brief_pytorch_amd.fit import Fitter
L, F = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(0)
dims = (64, 64, 64)
m = SIREN(features=F, layers=L, w0=20).to('cuda')
tv = torch.rand(64 ** 3, 1, device='cuda') * 100
fit = Fitter(m, tv, dims, sampler='full', sample_size=0)
for _ in range(int(os.environ.get("STAMP_STEPS", "2000"))): fit.step()
torch.cuda.synchronize()
nrec = 512 * 4
rec = m._ws[:nrec * 1056].view(nrec, 1056).cpu().numpy()      # k_small path: the record region starts the workspace
st = rec[:, 1030:1040]
names = ['inputs+layer0', 'fwd chain', 'fwd epilogue (sin, image)', 'head+loss+head grads', 'top delta + recompute + transposes + image',
         'dgrad chain', 'wgrad MFMAs + db', 'dl, first-layer grads, tile end']
tot = st.sum(1)
print('waves with stamps:', (tot > 0).sum(), 'mean total cycles/wave: %.0f' % tot[tot > 0].mean())
for i, nme in enumerate(names):
    v = st[tot > 0, i]
    print('%-46s mean %9.0f cycles  %5.1f%%   (min %9.0f max %9.0f)' % (nme, v.mean(), 100 * v.mean() / tot[tot > 0].mean(), v.min(), v.max()))
