"""Size-independent properties at BASELINE.json's full sizes (config 2: 256^3 volume, 4x256 SIREN,
100 000 samples per step), where the CPU oracle would take minutes: bitwise determinism,
linearity of the gradient under batch splitting, decode chunk invariance, fused de-normalise ==
oracle epilogue on the kernel's own output, GPU SSE / SSIM == numpy on a slab."""
import numpy as np
import pytest
import torch

from brief_pytorch_amd import _lib
from brief_pytorch_amd.fit import Fitter
from brief_pytorch_amd.metrics import gpu_ssim_u16
from brief_pytorch_amd.networks import SIREN
from brief_pytorch_amd.synthetic import make_volume_torch
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DIMS = (256, 256, 256)
N = 100000


@pytest.fixture(scope="module")
def volume():
    vol = make_volume_torch(DIMS, seed=7, device="cuda")
    t = vol.view(-1, 1).to(torch.float32)
    vmin, vmax = float(t.min().item()), float(t.max().item())
    tgt = (t - np.float32(vmin)) / np.float32(vmax - vmin)
    tgt *= np.float32(100.0)
    return vol, tgt, vmin, vmax


def _net(seed=42):
    torch.manual_seed(seed)
    return SIREN(features=256, layers=5, w0=20).to("cuda")


def test_fit_is_bitwise_deterministic_at_full_size(volume):
    vol, tgt, _, _ = volume
    out = []
    for _ in range(2):
        m = _net()
        f = Fitter(m, tgt, DIMS, sampler="randompoint", sample_size=N, seed=5)
        for _ in range(3):
            loss = f.step()
        out.append((m.params.clone(), loss.clone(), f.s2.clone()))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])
    assert torch.isfinite(out[0][0]).all()


def test_gradient_is_linear_in_the_batch_at_full_size(volume):
    vol, tgt, _, _ = volume
    m = _net()
    idx = torch.randint(0, tgt.shape[0], (N,), device="cuda")
    l_all, _ = m.train_step(N, tgt, idx=idx, grid=(DIMS, -1.0, 1.0))
    g_all, l_all = m.grads.double().clone(), l_all.item()
    h = 43210
    la, _ = m.train_step(h, tgt, idx=idx[:h].contiguous(), grid=(DIMS, -1.0, 1.0))
    ga, la = m.grads.double().clone(), la.item()
    lb, _ = m.train_step(N - h, tgt, idx=idx[h:].contiguous(), grid=(DIMS, -1.0, 1.0))
    gb, lb = m.grads.double().clone(), lb.item()
    comb = (ga * h + gb * (N - h)) / N
    assert float((comb - g_all).abs().max() / g_all.abs().max()) < 2e-5
    assert abs((la * h + lb * (N - h)) / N - l_all) / l_all < 1e-5


def test_decode_properties_at_full_size(volume):
    vol, tgt, vmin, vmax = volume
    m = _net(3)
    f = Fitter(m, tgt, DIMS, sampler="randompoint", sample_size=N, seed=1)
    for _ in range(20):
        f.step()
    total = int(np.prod(DIMS))
    whole = m.decode_grid(DIMS)                                    # 16.8 M voxels
    off, cnt = 5_000_017, 1_234_567                                # ragged chunk
    assert torch.equal(m.decode_grid(DIMS, offset=off, count=cnt).view(-1), whole.view(-1)[off:off + cnt])
    u16 = m.decode_grid(DIMS, out_kind="u16", scale=(0.0, 100.0), vrange=(vmin, vmax))
    side = {"dtype": "uint16", "min": vmin, "max": vmax}
    sl = slice(7_000_000, 9_000_000)
    assert np.array_equal(u16.view(-1)[sl].cpu().numpy(), O.invnormalize(whole.view(-1)[sl].cpu().numpy(), side).ravel())
    # last voxels of the grid (tile tail) and the first
    assert torch.equal(m.decode_grid(DIMS, offset=total - 5, count=5).view(-1), whole.view(-1)[-5:])
    # GPU SSE and SSIM against numpy on a slab of the volume
    dec = u16.view(DIMS)
    orig = vol.view(DIMS)
    sse = torch.zeros(1, dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib().brief_sse_u16(_lib.ptr(orig), _lib.ptr(dec), total, _lib.ptr(sse), _lib.stream_ptr()))
    a, b = orig[:6].cpu().numpy(), dec[:6].cpu().numpy()
    d6 = a.astype(np.int64) - b.astype(np.int64)
    sse6 = torch.zeros(1, dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib().brief_sse_u16(_lib.ptr(orig), _lib.ptr(dec), a.size, _lib.ptr(sse6), _lib.stream_ptr()))
    assert sse6.item() == float((d6 * d6).sum()) and sse.item() >= sse6.item()
    s, n = gpu_ssim_u16(orig[:6].contiguous(), dec[:6].contiguous())
    assert abs(s / n - O.ssim(a[..., None].astype(np.float32), b[..., None].astype(np.float32), 65535)) < 5e-5
