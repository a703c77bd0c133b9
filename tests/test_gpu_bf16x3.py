"""BRIEF_PREC_BF16X3 (`precision="bf16x3"`): split-precision hidden GEMMs — every operand a hi + lo pair of 16-bit halves (fp16 in the
forward chains, bf16 in the backward chains and the weight-gradient GEMM), three 16-bit MFMAs per product — held to the SAME bands as the fp32 path (forward 2e-5 of max|y|, loss 1e-5, every gradient tensor 1e-4 of its max-abs,
loss traces 1e-4 against the reference's goldens / the oracle), which the bf16 mode (bands 1e-2) is not.  Never the default and not
bit-identical to fp32; inference runs the same forward chains (fp16 halves), ~2e-6 from the f32 kernel."""
import ctypes as C

import os

import numpy as np
import pytest
import torch

from brief_pytorch_amd import _lib
from brief_pytorch_amd.fit import Fitter, MultiFitter
from brief_pytorch_amd.networks import SIREN
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def _net(L, F, w0=20.0, cin=3, cout=1, seed=0, precision="bf16x3", ws=None, bs=None):
    torch.manual_seed(seed)
    m = SIREN(coords_channel=cin, data_channel=cout, features=F, layers=L, w0=w0, precision=precision)
    if ws is not None:
        for l in range(L):
            m.net[l][0].weight.data = torch.from_numpy(ws[l])
            m.net[l][0].bias.data = torch.from_numpy(bs[l])
    p = m.params.numpy().copy()
    return m.to(DEV), O.make_desc(cin, cout, L, F, w0), p


@pytest.mark.parametrize("L,F,cin,cout,n,use_w,loss", [(5, 256, 3, 1, 20000, False, "datal2"), (4, 128, 3, 1, 5000, True, "datal2"),
                                                      (3, 200, 2, 3, 3333, False, "datasmoothl1"), (6, 96, 3, 1, 1000, True, "datal2"),
                                                      (9, 160, 3, 2, 2049, False, "datal2"), (2, 70, 3, 1, 31, False, "datal2"),
                                                      (5, 33, 3, 1, 700, True, "datasmoothl1")])
def test_train_step_meets_the_fp32_bands(L, F, cin, cout, n, use_w, loss):
    m, d, p = _net(L, F, cin=cin, cout=cout, seed=L * 100 + F)
    rng = np.random.default_rng(F + n)
    x = rng.uniform(-1, 1, size=(n, cin)).astype(np.float32)
    y = rng.uniform(0, 100, size=(n, cout)).astype(np.float32)
    w = np.where(rng.uniform(size=(n, cout)) < 0.5, 0.25, 1.0).astype(np.float32) if use_w else None
    kind, beta, thr = (1, 0.5, 30.0) if loss == "datasmoothl1" else (0, 0.01, 0.0)
    l1, yh = m.train_step(n, torch.from_numpy(y).to(DEV), coords=torch.from_numpy(x).to(DEV), weights=torch.from_numpy(w).to(DEV) if use_w else None,
                          loss=loss, thr=thr, beta=beta, want_yhat=True)
    g1 = m.grads.clone()
    lo, go, yo, _ = O.loss_grad(d, p, x, y, w, kind, thr, beta)
    assert relerr(yh.cpu().numpy(), yo) < 2e-5
    assert abs(l1.item() - lo) / abs(lo) < 1e-5
    gw, gb = O.unpack_params(d, go)
    mw, mb = O.unpack_params(d, m.grads.cpu().numpy())
    worst = max(max(relerr(mw[k], gw[k]), relerr(mb[k], gb[k])) for k in range(L))
    assert worst < 1e-4, worst
    l2, _ = m.train_step(n, torch.from_numpy(y).to(DEV), coords=torch.from_numpy(x).to(DEV), weights=torch.from_numpy(w).to(DEV) if use_w else None,
                         loss=loss, thr=thr, beta=beta)
    assert torch.equal(g1, m.grads) and l1.item() == l2.item()          # fixed summation order: bit-reproducible


def test_random_configurations_vs_oracle_at_the_fp32_bands():
    """the fp32 path's seeded random walk (tests/test_gpu_parity.py::test_random_configurations_vs_oracle: layers 2..11, widths 1..256
    around every tile boundary, cin 2/3, cout 1..4, sine head, both losses, weight maps, thresholds, ragged batches) on the split-
    precision kernels: the TRAIN kernel's yhat <= 2e-5, loss <= 1e-5, every gradient tensor <= 1e-4 — or 3x the oracle's own
    f32-vs-f64 distance where that is larger (deep one- and two-wide nets; tools/fuzz_parity.py N seed bf16x3 runs it for longer:
    400 configurations, four failures, all gradients of 1- and 2-wide nets 6 to 11 layers deep)"""
    rng = np.random.default_rng(2024)
    widths = list(range(1, 65)) + [65, 95, 96, 97, 127, 128, 129, 160, 191, 192, 200, 223, 224, 255, 256]
    worst_f, worst_g = 0.0, 0.0
    for case in range(60):
        L = int(rng.integers(2, 12))
        F = int(rng.choice(widths))
        cin, cout = int(rng.choice([2, 3])), int(rng.choice([1, 1, 1, 2, 3, 4]))
        oa = bool(rng.random() < 0.15)
        w0 = float(rng.choice([10.0, 20.0, 30.0]))
        n = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 255, 257, 1000, 2049, int(rng.integers(1, 6000))]))
        kind, thr, beta = int(rng.integers(0, 2)), float(rng.choice([0.0, 30.0, 200.0])), float(rng.choice([0.01, 1.0, 20.0]))
        use_w = bool(rng.random() < 0.6)
        torch.manual_seed(case)
        m = SIREN(coords_channel=cin, data_channel=cout, features=F, layers=L, w0=w0, output_act=oa, precision="bf16x3")
        d = O.make_desc(cin, cout, L, F, w0, 30.0, oa)
        p = m.params.numpy().copy()
        m.to(DEV)
        x = rng.uniform(-1, 1, size=(n, cin)).astype(np.float32)
        y = (rng.uniform(-1, 1, size=(n, cout)) if oa else rng.uniform(0, 100, size=(n, cout))).astype(np.float32)
        w = np.where(rng.uniform(size=(n, cout)) < 0.5, 0.25, 1.0).astype(np.float32) if use_w else np.ones((n, cout), np.float32)
        tag = (case, L, F, cin, cout, oa, w0, n, kind, thr, beta, use_w)
        loss, yh = m.train_step(n, torch.from_numpy(y).to(DEV), coords=torch.from_numpy(x).to(DEV), weights=torch.from_numpy(w).to(DEV) if use_w else None,
                                loss=["datal2", "datasmoothl1"][kind], thr=thr, beta=beta, want_yhat=True)
        lo, go, yo, _ = O.loss_grad(d, p, x, y, w, kind, thr, beta)
        _, go64, _, _ = O.loss_grad(d, p, x, y, w, kind, thr, beta, f64=True)
        e_f = relerr(yh.cpu().numpy(), yo)
        assert e_f < 2e-5, tag
        assert abs(loss.item() - lo) <= 1e-5 * max(abs(lo), 1e-2 * float(np.mean(y.astype(np.float64) ** 2))), tag
        gw, gb = O.unpack_params(d, go)
        gw64, gb64 = O.unpack_params(d, go64)
        band = max(1e-4, 3.0 * max(max(relerr(gw[l], gw64[l]), relerr(gb[l], gb64[l])) for l in range(L)))
        mw, mb = O.unpack_params(d, m.grads.cpu().numpy())
        gmax = max(float(np.max(np.abs(t))) for t in list(gw) + list(gb))
        for l in range(L):
            for got, ref in ((mw[l], gw[l]), (mb[l], gb[l])):
                scale = max(float(np.max(np.abs(ref))), 1e-6 * gmax, 1e-30)
                e_g = float(np.max(np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)))) / scale
                assert e_g < band, (tag, l, e_g, band)
                worst_g = max(worst_g, e_g / (band / 1e-4))
        worst_f = max(worst_f, e_f)
    print("bf16x3 random walk, 60 configurations: worst yhat %.2e (band 2e-5), worst gradient tensor %.2e of a 1e-4 band" % (worst_f, worst_g))


def test_forward_and_decode_on_the_split_precision_kernel():
    """inference (forward / decode_grid) of a bf16x3 net runs the forward chains on fp16 halves: against the oracle inside the fp32
    band (2e-5; measured ~2e-6, the f32 kernel's own distance), within 5e-6 of the f32 kernel, bit-reproducible; the fused
    de-normalise + truncating uint16 cast then differs from the f32 kernel's by at most one code, in a small share of the voxels"""
    a, d, p = _net(5, 200, seed=3, precision="bf16x3")
    b, _, _ = _net(5, 200, seed=3, precision="fp32")
    xh = np.random.default_rng(1).uniform(-1, 1, size=(7777, 3)).astype(np.float32)
    x = torch.from_numpy(xh).to(DEV)
    ya, yb = a.forward(x), b.forward(x)
    yo = O.forward(d, p, xh)
    ea, eb = relerr(ya.cpu().numpy(), yo), relerr(yb.cpu().numpy(), yo)
    print("forward against the oracle: bf16x3 %.2e, fp32 %.2e; against each other %.2e" % (ea, eb, relerr(ya.cpu().numpy(), yb.cpu().numpy())))
    assert ea < 2e-5 and relerr(ya.cpu().numpy(), yb.cpu().numpy()) < 5e-6
    assert torch.equal(ya, a.forward(x))
    dims = (40, 50, 60)
    fa, fb = a.decode_grid(dims), b.decode_grid(dims)
    assert relerr(fa.cpu().numpy(), O.forward(d, p, O.grid_coords(dims))) < 2e-5 and relerr(fa.cpu().numpy(), fb.cpu().numpy()) < 5e-6
    lo, hi = float(fb.min()), float(fb.max())
    ua = a.decode_grid(dims, out_kind="u16", scale=(lo, hi), vrange=(0.0, 65535.0)).cpu().numpy().astype(np.int64)
    ub = b.decode_grid(dims, out_kind="u16", scale=(lo, hi), vrange=(0.0, 65535.0)).cpu().numpy().astype(np.int64)
    diff = np.abs(ua - ub)
    print("uint16 decode over the net's full range: %.1f %% of the voxels differ, by at most %d code(s)" % (100.0 * np.mean(diff > 0), diff.max()))
    assert diff.max() <= 1 and np.mean(diff > 0) < 0.25


def test_trace_against_the_reference_golden_and_fit_loop_identities(golden):
    """the 3000-step golden's net (5x128, full batch of 30 720 samples) for its first 50 steps against the REFERENCE's own loss
    trace at the fp32 band (1e-4), through brief_siren_fit; run(k) == k x step() and co-trained == alone, bit for bit"""
    g = golden("half")
    L, F, w0, _ = (int(v) for v in g["cfg"])
    vol = g["vol"]
    vn, side = O.normalize(vol)
    thr = float(O.normalize(np.array([65535], np.uint16), vmin=side["min"], vmax=side["max"])[0][0])
    tv = torch.from_numpy(vn.reshape(-1, 1)).to(DEV)
    fits = []
    for _ in range(3):
        m, _, _ = _net(L, F, w0=float(w0), ws=[g["init_w%d" % l] for l in range(L)], bs=[g["init_b%d" % l] for l in range(L)])
        fits.append(Fitter(m, tv, vol.shape[:3], sampler="full", optimizer="Adamax", lr=1e-3, thr=thr,
                           scheduler={"name": "MultiStepLR", "milestones": [50000, 60000, 70000], "gamma": 0.2}))
    trace = fits[0].run(50, log=True).cpu().numpy().astype(np.float64)
    ref = g["f32_losses"][:50]
    err = np.abs(trace - ref) / ref
    print("bf16x3 50-step trace against the reference's run: max relative error %.2e" % err.max())
    assert err.max() < 1e-4
    stepped = np.asarray([float(fits[1].step()) for _ in range(50)], np.float32)
    assert np.array_equal(stepped, trace.astype(np.float32)) and torch.equal(fits[0].m.params, fits[1].m.params) and torch.equal(fits[0].m.packed, fits[1].m.packed)
    other, _, _ = _net(4, 22, seed=8, precision="fp32")
    tv2 = (torch.rand(16 ** 3, 1, generator=torch.Generator().manual_seed(2)) * 100).to(DEV)
    solo = Fitter(other, tv2, (16, 16, 16), sampler="full")
    solo2, _, _ = _net(4, 22, seed=8, precision="fp32")
    grp = MultiFitter([fits[2], Fitter(solo2, tv2, (16, 16, 16), sampler="full")])
    grp.run(50)
    solo.run(50)
    assert torch.equal(fits[2].m.params, fits[0].m.params) and torch.equal(solo2.params, other.params)


def test_full_size_step_and_trace_of_the_headline_shape():
    """BASELINE config 2's shape (4x256, 100 000 randompoint samples of a 256^3 volume): one step against the oracle at the
    fp32 bands, then 12 optimizer steps against the oracle's loop (trace 1e-4)"""
    from brief_pytorch_amd.synthetic import make_volume_torch
    dims, L, F, N = (256, 256, 256), 5, 256, 100000
    vol = make_volume_torch(dims, seed=11, device=DEV)
    t = vol.view(-1, 1).to(torch.float32)
    tgt = (t - t.min()) / (t.max() - t.min()) * 100.0
    m, d, p0 = _net(L, F, seed=12)
    pop = tgt.shape[0]
    idx = torch.empty(N, dtype=torch.int64, device=DEV)
    _lib.check(_lib.lib().brief_sample_indices(_lib.ptr(idx), N, pop, 77, 1, _lib.stream_ptr()))
    loss, yh = m.train_step(N, tgt, idx=idx, grid=(dims, -1.0, 1.0), want_yhat=True)
    ih = idx.cpu().numpy()
    tgt_h = tgt.cpu().numpy()
    lo, go, yo, _ = O.loss_grad(d, p0, O.grid_coords(dims, idx=ih), tgt_h[ih])
    gw, gb = O.unpack_params(d, go)
    mw, mb = O.unpack_params(d, m.grads.cpu().numpy())
    worst = max(max(relerr(mw[k], gw[k]), relerr(mb[k], gb[k])) for k in range(L))
    print("bf16x3 full-size step: yhat %.2e loss %.2e worst gradient tensor %.2e" % (relerr(yh.cpu().numpy(), yo), abs(loss.item() - lo) / lo, worst))
    assert relerr(yh.cpu().numpy(), yo) < 2e-5 and abs(loss.item() - lo) / lo < 1e-5 and worst < 1e-4
    steps = 12
    fit = Fitter(m, tgt, dims, sampler="randompoint", sample_size=N, optimizer="Adamax", lr=1e-3, seed=77)
    trace = fit.run(steps, log=True).cpu().numpy().astype(np.float64)
    p, s1, s2 = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    ref = []
    for k in range(1, steps + 1):
        _lib.check(_lib.lib().brief_sample_indices(_lib.ptr(idx), N, pop, 77, k, _lib.stream_ptr()))
        ih = idx.cpu().numpy()
        l_k, g_k, _, _ = O.loss_grad(d, p, O.grid_coords(dims, idx=ih), tgt_h[ih])
        O.optim_step("Adamax", p, g_k, s1, s2, 1e-3, k)
        ref.append(l_k)
    err = np.abs(trace - np.asarray(ref)) / np.asarray(ref)
    print("bf16x3 full-size trace over %d steps: max relative error %.2e" % (steps, err.max()))
    assert err.max() < 1e-4



def test_singletask_bf16x3_option_matches_the_fp32_run(tmp_path):
    """Compress.precision: bf16x3 through NFGR.compress: fp32 weight files, side info records the precision, the stored artefact
    decodes to the stored volume; the 150-step fit lands where the fp32 fit of the same seed lands (PSNR within 0.05 dB: the two
    paths agree to ~1e-5 per step, the fit has not had time to diverge)"""
    from brief_pytorch_amd import config
    from brief_pytorch_amd.framework import NFGR, MyLogger
    from brief_pytorch_amd.synthetic import make_volume
    from brief_pytorch_amd.tool import read_img, save_img
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    vol = make_volume((16, 32, 32), seed=3)
    path = str(tmp_path / "v.tif")
    save_img(path, vol)
    psnr = {}
    for prec in ("fp32", "bf16x3"):
        opt = config.load(os.path.join(root, "opt", "SingleTask", "default.yaml"))
        cf = opt.CompressFramework
        cf.Compress.max_steps = 150
        cf.Compress.checkpoints = "none"
        cf.Compress.param.filesize_ratio = 0
        cf.Compress.param.given_size = 4.0 * SIREN.calc_param_count(3, 1, 160, 4)
        cf.Module.phi.layers = 4
        cf.Compress.precision = prec
        opt.Log.outputs_dir = str(tmp_path / ("outputs_" + prec))
        opt.Log.time = False
        Log = MyLogger(**opt.Log)
        torch.manual_seed(42)
        fw = NFGR(cf, Log=Log)
        res = fw.compress(path)
        psnr[prec] = res[150]["psnr"]
        if prec == "bf16x3":
            sdir = os.path.join(Log.logdir, "steps150")
            side = config.load(os.path.join(sdir, "compressed", "sideinfos.yaml"))
            assert side["phi_precision"] == "bf16x3" and side["phi_features"] == 160
            assert os.path.getsize(os.path.join(sdir, "compressed", "module", "weight-1-160-160")) == 160 * 160 * 4
            again = NFGR.decompress(config.to_opt({"CompressFramework": cf}), os.path.join(sdir, "compressed", "module"), dict(side))
            assert np.array_equal(again, read_img(os.path.join(sdir, "decompressed", "v_decompressed.tif")))
    print("150-step SingleTask fit: fp32 %.3f dB, bf16x3 %.3f dB" % (psnr["fp32"], psnr["bf16x3"]))
    assert psnr["fp32"] > 25 and abs(psnr["fp32"] - psnr["bf16x3"]) < 0.05


def test_autograd_route_external_gradients():
    """the reference-style loop (forward -> torch loss -> backward) on a bf16x3 module: autograd's dL/dyhat enters the fused backward as
    BRIEF_LOSS_EXTERNAL; the parameter gradient agrees with the fp32 module's to the fp32 band (every tensor 1e-4 of its max-abs)"""
    import torch.nn.functional as Fnn
    grads = {}
    dims = (12, 16, 20)
    n = 12 * 16 * 20
    y = (torch.rand(n, 1, generator=torch.Generator().manual_seed(1)) * 100).to(DEV)
    lin = [torch.linspace(-1, 1, d) for d in dims]
    x = torch.stack(torch.meshgrid(*lin, indexing="ij"), -1).reshape(1, *dims, 3).to(DEV)
    for prec in ("fp32", "bf16x3"):
        torch.manual_seed(7)
        m = SIREN(features=200, layers=5, w0=20, precision=prec).to(DEV)
        m.requires_grad_(True)
        yhat = m.forward(x)
        assert yhat.requires_grad
        Fnn.mse_loss(yhat, y.view(1, *dims, 1)).backward()
        grads[prec] = m.params.grad.detach().cpu().numpy().copy()
        d = O.make_desc(3, 1, 5, 200, 20.0)
    a, b = O.unpack_params(d, grads["bf16x3"]), O.unpack_params(d, grads["fp32"])
    worst = max(max(relerr(a[0][k], b[0][k]), relerr(a[1][k], b[1][k])) for k in range(5))
    print("autograd route: worst gradient tensor, bf16x3 against fp32: %.2e" % worst)
    assert worst < 1e-4


def test_dividetask_bf16x3_blocks(tmp_path):
    """Compress.precision: bf16x3 through NFGR.compress_divide: four 16x16x16 blocks with ~150-wide nets co-trained by brief_multi_fit,
    the merged volume rebuilt from the stored artefact tree bit for bit, PSNR where the fp32 run of the same seed lands"""
    from brief_pytorch_amd import config, misc
    from brief_pytorch_amd.framework import NFGR, MyLogger
    from brief_pytorch_amd.synthetic import make_volume
    from brief_pytorch_amd.tool import read_img, save_img
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    vol = make_volume((16, 32, 32), seed=3)
    path = str(tmp_path / "blk.tif")
    save_img(path, vol)
    psnr = {}
    for prec in ("fp32", "bf16x3"):
        opt = config.load(os.path.join(root, "opt", "SingleTask", "default.yaml"))
        cf = opt.CompressFramework
        cf.Compress.max_steps, cf.Compress.checkpoints, cf.Compress.loss_log_freq = 150, "none", 50
        cf.Compress.param.filesize_ratio, cf.Compress.param.given_size = 0, 4 * 4.0 * SIREN.calc_param_count(3, 1, 150, 4)
        cf.Compress.divide.divide_type, cf.Compress.divide.param_alloc = "total_1_2_2", "by_size"
        cf.Module.phi.layers = 4
        cf.Compress.precision = prec
        opt.Log.outputs_dir, opt.Log.time = str(tmp_path / ("outputs_" + prec)), False
        Log = MyLogger(**opt.Log)
        torch.manual_seed(42)
        fw = NFGR(cf, Log=Log)
        res = fw.compress_divide(path, opt)
        psnr[prec] = res[150]["psnr"]
        if prec == "bf16x3":
            cdir = os.path.join(Log.logdir, "steps150", "compressed")
            names = sorted(os.listdir(os.path.join(cdir, "module")))
            assert names == sorted(c["name"] for c in misc.divide_data(vol, "total_1_2_2")[0])
            side = config.load(os.path.join(cdir, "sideinfos", names[0], "sideinfos.yaml"))
            assert side["phi_precision"] == "bf16x3" and 128 < side["phi_features"] <= 256
            merged = read_img(os.path.join(Log.logdir, "steps150", "decompressed", "blk_decompressed.tif"))
            again = fw.decompress_divide(os.path.join(cdir, "sideinfos.yaml"), os.path.join(cdir, "module"), os.path.join(cdir, "sideinfos"))
            assert np.array_equal(again, merged)
    print("DivideTask 150 steps: fp32 %.3f dB, bf16x3 %.3f dB" % (psnr["fp32"], psnr["bf16x3"]))
    assert psnr["fp32"] > 24 and abs(psnr["fp32"] - psnr["bf16x3"]) < 0.1
