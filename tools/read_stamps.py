"""scratch: read the cycle stamps of the diagnostic k_fused build"""
import shutil, sys
import os; os.environ.setdefault("BRIEF_LIB", os.path.abspath("brief_pytorch_amd/libbrief_hip_stamps.so"))
import torch, numpy as np
sys.path.insert(0, '.')
from brief_pytorch_amd import _lib
from brief_pytorch_amd.networks import SIREN
from brief_pytorch_amd.fit import Fitter
torch.manual_seed(0)
pop = 256**3
m = SIREN(features=256, layers=5, w0=20, precision=os.environ.get('STAMP_PREC', 'fp32')).to('cuda')
tv = torch.rand(pop, 1, device='cuda') * 100
fit = Fitter(m, tv, (256,256,256), sample_size=100000)
for _ in range(int(os.environ.get("STAMP_STEPS", "600"))): fit.step()      # steady state: the first steps after idle run at ~1.85 GHz
torch.cuda.synchronize()
FP, npad, hidden = 256, 100000, 3
rec_off = 2 * hidden * FP * npad
nrec = 512 * 4
rec = m._ws[rec_off:rec_off + nrec * 1056].view(nrec, 1056).cpu().numpy()
st = rec[:, 1030:1040]
ck = rec[:, 1040:1042]
ok = ck[:, 1] > 0
print('in-kernel clock: %.3f GHz (shader cycles per 100 MHz reference tick, median over waves); kernel lifetime %.1f us' % (float(np.median(ck[ok, 0] / ck[ok, 1])) * 0.1, float(np.median(ck[ok, 1])) / 100.0))
names = ['inputs+layer0', 'fwd chain', 'barrier after chain', 'fwd epilogue(stash,sincos,image)', 'barrier after image', 'head+loss',
         'head-grad + delta + d0 transposes', 'D store + z prefetch', 'barriers+image before bwd chain', 'bwd chain']
tot = st.sum(1)
print('waves with stamps:', (tot > 0).sum(), 'mean total cycles/wave: %.0f' % tot[tot > 0].mean())
for i, nme in enumerate(names):
    v = st[tot > 0, i]
    print('%-36s mean %9.0f cycles  %5.1f%%   (min %9.0f max %9.0f)' % (nme, v.mean(), 100 * v.mean() / tot[tot > 0].mean(), v.min(), v.max()))
# ---- wgrad stamps
rec_floats = 256 * 8 * 4 * 1056
ws = m._ws[rec_off + rec_floats - 256 * 8 * 8: rec_off + rec_floats].view(256, 8, 8).cpu().numpy()
w = ws[:255].reshape(-1, 8)[:, :4]
tot = w.sum(1)
print('wgrad: mean total cycles/wave %.0f (chunks per split ~36.8)' % tot.mean())
for i, nme in enumerate(['stage A panel', 'MFMA groups (+B staging)', 'issue next loads', 'barrier']):
    print('%-28s mean %9.0f  %5.1f%%  per chunk %7.0f' % (nme, w[:, i].mean(), 100 * w[:, i].mean() / tot.mean(), w[:, i].mean() / 36.8))
