"""seed-to-seed spread of the end-of-fit PSNR, fused path against plain PyTorch on the GPU (see wide_dynamics_check.py):
    python tools/wide_dynamics_seeds.py F steps edge nseeds"""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from brief_pytorch_amd.fit import Fitter
from brief_pytorch_amd.networks import SIREN
from brief_pytorch_amd.synthetic import make_volume_torch

F, steps, E, ns = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
n, L, w0 = 100000, 5, 20.0
dims, vox = (E, E, E), E ** 3
vol = make_volume_torch(dims, seed=42, detail=64)
vf = vol.to(torch.int32).to(torch.float32)
vmin, vmax = float(vf.min()), float(vf.max())
tv = ((vf - vmin) / (vmax - vmin) * 100.0).reshape(vox, 1).contiguous()
lin = torch.linspace(-1, 1, E, device="cuda")

def psnr_of(dec):
    d = (dec.clamp(0, 100) / 100.0 * (vmax - vmin) + vmin).to(torch.float64) - vol.reshape(-1).to(torch.float64)
    return -10.0 * np.log10(float((d * d).mean()) / 65535.0 ** 2)

def fused(seed, use_torch_idx):
    torch.manual_seed(42)
    m = SIREN(coords_channel=3, data_channel=1, features=F, layers=L, w0=w0).to("cuda")
    if not use_torch_idx:
        Fitter(m, tv, dims, sampler="randompoint", sample_size=n, seed=seed, lr=1e-3).run(steps)
    else:      # the fused step on torch.randint's indices (what the plain path draws)
        fit = Fitter(m, tv, dims, sampler="randompoint", sample_size=n, seed=seed, lr=1e-3)
        g = torch.Generator(device="cuda").manual_seed(seed)
        for t in range(1, steps + 1):
            idx = torch.randint(0, vox, (n,), device="cuda", generator=g)
            m.fit_step(n, tv, fit.opt, fit.s1, fit.s2, 1e-3, t, idx=idx, grid=(dims, -1.0, 1.0))
    return psnr_of(m.decode_grid(dims).reshape(-1))

def plain(seed):
    torch.manual_seed(42)
    m = SIREN(coords_channel=3, data_channel=1, features=F, layers=L, w0=w0).to("cuda")
    ws = [m.net[l][0].weight.data.clone().requires_grad_(True) for l in range(L)]
    bs = [m.net[l][0].bias.data.clone().requires_grad_(True) for l in range(L)]
    opt = torch.optim.Adamax(ws + bs, lr=1e-3)
    def fwd(i):
        h = torch.stack([lin[i // (E * E)], lin[(i // E) % E], lin[i % E]], -1)
        for l in range(L - 1):
            h = torch.sin((w0 if l == 0 else 30.0) * (h @ ws[l].t() + bs[l]))
        return h @ ws[L - 1].t() + bs[L - 1]
    g = torch.Generator(device="cuda").manual_seed(seed)
    for _ in range(steps):
        idx = torch.randint(0, vox, (n,), device="cuda", generator=g)
        loss = ((fwd(idx) - tv[idx]) ** 2).mean()
        opt.zero_grad(); loss.backward(); opt.step()
    with torch.no_grad():
        return psnr_of(torch.cat([fwd(i).reshape(-1) for i in torch.arange(vox, device="cuda").split(1 << 19)]))

print("4x%d, %d^3, %d steps, PSNR dB per seed" % (F, E, steps))
print("  fused (Philox indices)      : %s" % " ".join("%.2f" % fused(40 + k, False) for k in range(ns)), flush=True)
print("  fused (torch.randint idx)   : %s" % " ".join("%.2f" % fused(40 + k, True) for k in range(ns)), flush=True)
print("  plain PyTorch               : %s" % " ".join("%.2f" % plain(40 + k) for k in range(ns)), flush=True)
