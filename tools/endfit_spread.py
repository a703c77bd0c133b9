"""spread of the 3000-step end-of-fit PSNR (tests/golden/half.npz case) under one-ulp perturbations of one initial weight:
   how much of a PSNR difference between two arithmetic variants is trajectory noise?   BRIEF_LIB=... python tools/endfit_spread.py [runs]"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from oracle import oracle as O
from brief_pytorch_amd.networks import SIREN
from brief_pytorch_amd.fit import Fitter
g = np.load('tests/golden/half.npz')
L, F, w0, steps = (int(v) for v in g["cfg"])
vol = g["vol"]
vn, side = O.normalize(vol)
thr = float(O.normalize(np.array([65535], np.uint16), vmin=side["min"], vmax=side["max"])[0][0])
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(0)
out = []
for k in range(runs):
    m = SIREN(features=F, layers=L, w0=w0)
    ws = [g["init_w%d" % l].copy() for l in range(L)]
    if k > 0:
        l = int(rng.integers(0, L)); w = ws[l].ravel(); i = int(rng.integers(0, w.size))
        w[i] = np.nextafter(w[i], np.float32(1e9) if rng.random() < 0.5 else np.float32(-1e9))
    for l in range(L):
        m.net[l][0].weight.data = torch.from_numpy(ws[l])
        m.net[l][0].bias.data = torch.from_numpy(g["init_b%d" % l])
    m.to('cuda')
    tv = torch.from_numpy(vn.reshape(-1, 1)).cuda()
    fit = Fitter(m, tv, vol.shape[:3], sampler="full", optimizer="Adamax", lr=1e-3, thr=thr,
                 scheduler={"name": "MultiStepLR", "milestones": [50000, 60000, 70000], "gamma": 0.2})
    losses = fit.run(steps, log=True).cpu().numpy().astype(np.float64)
    dec = m.decode_grid(vol.shape[:3], out_kind="u16", scale=(0.0, 100.0), vrange=(side["min"], side["max"])).cpu().numpy().reshape(vol.shape)
    ps = O.psnr(vol, dec, 65535)
    out.append(ps)
    print("run %d: loss at 1000/2000/2900/3000: %.5f %.5f %.5f %.5f  min over last 200: %.5f max: %.5f  PSNR %.3f dB" % (k, losses[999], losses[1999], losses[2899], losses[2999], losses[-200:].min(), losses[-200:].max(), ps), flush=True)
print("PSNR mean %.3f std %.3f min %.3f max %.3f (reference runs: %.3f, half_self %.3f)" % (np.mean(out), np.std(out), np.min(out), np.max(out), float(g["f32_psnr"][0]), float(np.load('tests/golden/half_self.npz')["f32_psnr"][0])))
