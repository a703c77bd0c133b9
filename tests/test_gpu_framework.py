"""End-to-end on the GPU through the reference-shaped framework API: SingleTask fit ->
artefact tree -> decode from the files -> metrics, against the reference's own 300-step run
(golden decode.npz) with the end-of-fit band of SURVEY.md Appendix F (|dPSNR| <= 0.1 dB), and
DivideTask on one rank."""
import copy
import os

import numpy as np
import pytest
import torch

from brief_pytorch_amd import config, misc
from brief_pytorch_amd.framework import NFGR, MyLogger
from brief_pytorch_amd.networks import SIREN
from brief_pytorch_amd.tool import read_img, save_img

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _opt(tmp_path, steps, ckpt, given):
    opt = config.load(os.path.join(ROOT, "opt", "SingleTask", "default.yaml"))
    cf = opt.CompressFramework
    cf.Compress.max_steps = steps
    cf.Compress.checkpoints = ckpt
    cf.Compress.param.filesize_ratio = 0
    cf.Compress.param.given_size = given
    cf.Compress.loss_log_freq = 50
    opt.Log.outputs_dir = str(tmp_path / "outputs")
    opt.Log.time = False
    return opt


def test_singletask_matches_reference_run(tmp_path, golden):
    g = golden("decode")
    vol = g["vol"]
    path = str(tmp_path / "vol.tif")
    save_img(path, vol)
    assert np.array_equal(read_img(path), vol)
    opt = _opt(tmp_path, 300, "every_100", 4.0 * SIREN.calc_param_count(3, 1, 22, 5))
    Log = MyLogger(**opt.Log)
    torch.manual_seed(42)                                   # reproduc(seed 42), as in the golden run
    fw = NFGR(opt.CompressFramework, Log=Log)
    res = fw.compress(path)
    assert sorted(res) == [100, 200, 300]
    assert abs(res[300]["psnr"] - g["psnr"][0]) < 0.1          # reference: same init, same batches
    assert abs(res[300]["ssim"] - g["ssim"][0]) < 5e-3
    assert abs(res[300]["loss"] - g["losses"][-1]) / g["losses"][-1] < 5e-3
    # artefact tree
    sdir = os.path.join(Log.logdir, "steps300")
    mod = os.path.join(sdir, "compressed", "module")
    assert sorted(os.listdir(mod)) == sorted(["weight-0-22-3", "bias-0-22", "weight-1-22-22", "bias-1-22", "weight-2-22-22",
                                              "bias-2-22", "weight-3-22-22", "bias-3-22", "weight-4-1-22", "bias-4-1"])
    side = config.load(os.path.join(sdir, "compressed", "sideinfos.yaml"))
    assert set(side) == {"dtype", "min", "max", "normalized_min", "normalized_max", "data_shape", "phi_features", "phi_name"}
    assert side["phi_features"] == 22 and list(side["data_shape"]) == list(vol.shape)
    dec_file = read_img(os.path.join(sdir, "decompressed", "vol_decompressed.tif"))
    # decoding the stored artefact again reproduces the stored volume bit for bit
    again = NFGR.decompress(config.to_opt({"CompressFramework": opt.CompressFramework}), mod, dict(side))
    assert np.array_equal(again, dec_file)
    assert os.path.exists(os.path.join(Log.logdir, "performance.csv"))
    # main.py:429-438: projections of the original and of the decoded volume, in the data's format and as .png
    for tag, v in (("vol", vol), ("vol_decompressed", dec_file)):
        for ax, axis in zip("dhw", (0, 1, 2)):
            for e in (".tif", ".png"):
                assert np.array_equal(np.squeeze(read_img(os.path.join(sdir, "mip", "%s_mip_%s%s" % (tag, ax, e)))), v.max(axis)[..., 0]), (tag, ax, e)
    # the reference's final weights decode to (almost) the same volume as ours: same basin
    d = np.abs(dec_file.astype(np.int64) - g["dec_u16"].astype(np.int64))
    assert np.median(d) < 40


def test_randompoint_fit_reproduces_the_reference_trace_from_the_seed_alone(tmp_path, golden):
    """Compress.sampler.rng: torch: with the reference's seed and nothing else — no recorded index stream — the
    fit through NFGR.prepare_fit touches the voxels the reference's RandompointSampler touched (main.py:126-163, 653-661) and its
    50-step loss trace is the reference's to 1e-4, its final weights to 5e-5 (tests/golden/trace.npz: seed 42, 4 x 32, 1 000 samples)."""
    g = golden("trace")
    vol = g["pt_vol"]
    opt = _opt(tmp_path, 50, "none", 4.0 * SIREN.calc_param_count(3, 1, 32, 4))
    cf = opt.CompressFramework
    cf.Module.phi.layers = 4
    cf.Compress.sampler.name = "randompoint"
    cf.Compress.sampler.sample_size = 1000
    cf.Compress.sampler.rng = "torch"
    torch.manual_seed(42)                                   # reproduc(seed 42), as in the golden run
    fw = NFGR(cf, Log=None)
    ctx = fw.prepare_fit(str(tmp_path / "vol.tif"), data=vol, logdir=str(tmp_path))
    phi = ctx["phi"]
    for l in range(4):
        assert np.array_equal(phi.net[l][0].weight.data.cpu().numpy(), g["pt_init_w%d" % l])
    losses = ctx["fit"].run(50, log=True).cpu().numpy().astype(np.float64)
    assert np.max(np.abs(losses - g["pt_losses"]) / g["pt_losses"]) < 1e-4
    for l in range(4):
        assert np.max(np.abs(phi.net[l][0].weight.data.cpu().numpy() - g["pt_final_w%d" % l])) < 5e-5
    # the default (Philox, in-kernel) draws other voxels: same statistics, another trace
    cf.Compress.sampler.rng = "philox"
    torch.manual_seed(42)
    ctx2 = NFGR(cf, Log=None).prepare_fit(str(tmp_path / "vol.tif"), data=vol, logdir=str(tmp_path))
    other = ctx2["fit"].run(50, log=True).cpu().numpy().astype(np.float64)
    assert np.max(np.abs(other - g["pt_losses"]) / g["pt_losses"]) > 1e-3
    assert abs(other[-1] - g["pt_losses"][-1]) / g["pt_losses"][-1] < 0.5
    with pytest.raises(ValueError):
        cf.Compress.sampler.rng = "mt"
        NFGR(cf, Log=None).prepare_fit(str(tmp_path / "vol.tif"), data=vol, logdir=str(tmp_path))


@pytest.mark.parametrize("L,F", [(5, 256), (4, 40), (3, 300), (3, 600)])
def test_reference_loop_body_runs_on_the_module(L, F):
    """the reference's own loop body (main.py:385-400: zero_grad, forward, loss_func, backward, torch.optim step,
    scheduler) on the duck-typed module: after requires_grad_(True) forward() is differentiable w.r.t. the parameters
    (BRIEF_LOSS_EXTERNAL carries autograd's dL/dyhat into the fused backward) and torch.optim updates them in place."""
    import torch.nn.functional as Fnn
    from brief_pytorch_amd.fit import Fitter
    torch.manual_seed(7)
    a = SIREN(features=F, layers=L, w0=20).to("cuda")
    torch.manual_seed(7)
    b = SIREN(features=F, layers=L, w0=20).to("cuda")
    dims = (12, 16, 20)
    n = 12 * 16 * 20
    g = torch.Generator().manual_seed(1)
    y = (torch.rand(n, 1, generator=g) * 100).cuda()
    lin = [torch.linspace(-1, 1, d) for d in dims]
    x = torch.stack(torch.meshgrid(*lin, indexing="ij"), -1).reshape(1, *dims, 3).cuda()      # (1,d,h,w,3) as RandomCubeSampler yields
    # reference-style loop on a
    a.requires_grad_(True)
    opt = torch.optim.Adamax(a.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, [2, 3], 0.5)
    ref_losses = []
    for _ in range(5):
        opt.zero_grad()
        yhat = a.forward(x)
        assert yhat.shape == (1, *dims, 1) and yhat.requires_grad
        loss = Fnn.mse_loss(yhat, y.view(1, *dims, 1))
        loss.backward()
        if not ref_losses:
            g_first = a.params.grad.clone()
        opt.step()
        sched.step()
        ref_losses.append(float(loss.detach()))
    # fused path on b: same data, same schedule
    fit = Fitter(b, y, dims, sampler="full", optimizer="Adamax", lr=1e-3, scheduler={"name": "MultiStepLR", "milestones": [2, 3], "gamma": 0.5})
    b.train_step(n, y, grid=(dims, -1.0, 1.0))
    assert float((g_first - b.grads).abs().max()) <= 2e-5 * float(b.grads.abs().max())
    fused_losses = [float(fit.step()) for _ in range(5)]
    assert np.allclose(ref_losses, fused_losses, rtol=2e-5)
    assert float((a.params - b.params).abs().max()) < 2e-5
    with torch.no_grad():
        assert not a.forward(x).requires_grad


def test_singletask_2d_rgb_image(tmp_path):
    """2-D RGB inputs (coords_channel = 2, data_channel = 3, uint8) through NFGR.compress / decompress"""
    rng = np.random.default_rng(0)
    yy, xx = np.meshgrid(np.linspace(0, 1, 48), np.linspace(0, 1, 64), indexing="ij")
    img = np.stack([120 + 100 * np.sin(6 * xx + 2 * yy), 128 + 90 * np.cos(5 * yy), 100 + 80 * np.sin(4 * (xx + yy))], -1)
    img = np.clip(img + rng.normal(0, 2, img.shape), 0, 255).astype(np.uint8)
    path = str(tmp_path / "rgb.png")                      # utils/tool.py:85-91, 100-101: the .png route of read_img / save_img
    save_img(path, img)
    assert np.array_equal(read_img(path), img)
    opt = _opt(tmp_path, 4000, "none", 4.0 * SIREN.calc_param_count(2, 3, 48, 4))
    cf = opt.CompressFramework
    cf.Module.phi.coords_channel, cf.Module.phi.data_channel, cf.Module.phi.layers = 2, 3, 4
    cf.Compress.preprocess.clip = [0, 255]
    cf.Decompress.postprocess.clip = [0, 255]
    cf.Compress.loss.weight = ["value_255_255_1"]
    cf.Compress.loss.weight_thres = 255
    cf.Decompress.mip = False
    Log = MyLogger(**opt.Log)
    torch.manual_seed(42)
    fw = NFGR(cf, Log=Log)
    res = fw.compress(path)
    assert res[4000]["psnr"] > 20
    sdir = os.path.join(Log.logdir, "steps4000")
    side = config.load(os.path.join(sdir, "compressed", "sideinfos.yaml"))
    assert list(side["data_shape"]) == [48, 64, 3] and side["phi_features"] == 48 and side["dtype"] == "uint8"
    assert os.path.exists(os.path.join(sdir, "compressed", "module", "weight-0-48-2")) and os.path.exists(os.path.join(sdir, "compressed", "module", "weight-3-3-48"))
    dec = read_img(os.path.join(sdir, "decompressed", "rgb_decompressed.png"))
    assert dec.shape == img.shape and dec.dtype == np.uint8
    again = NFGR.decompress(config.to_opt({"CompressFramework": cf}), os.path.join(sdir, "compressed", "module"), dict(side))
    assert np.array_equal(again, dec)
    d = dec.astype(np.float64) - img.astype(np.float64)
    assert abs(-10 * np.log10((d * d).mean() / 255.0 ** 2) - res[4000]["psnr"]) < 1e-4       # cal_psnr works in float32


def test_dividetask_single_rank(tmp_path):
    from brief_pytorch_amd.synthetic import make_volume
    vol = make_volume((16, 32, 32), seed=3)
    path = str(tmp_path / "blk.tif")
    save_img(path, vol)
    opt = _opt(tmp_path, 200, "none", 20000.0)
    cf = opt.CompressFramework
    cf.Compress.divide.divide_type = "total_1_2_2"
    cf.Compress.divide.param_alloc = "by_size"
    cf.Module.phi.layers = 4
    Log = MyLogger(**opt.Log)
    torch.manual_seed(42)
    fw = NFGR(cf, Log=Log)
    res = fw.compress_divide(path, opt)
    assert list(res) == [200] and res[200]["psnr"] > 24     # 200 steps only: a sanity floor, not a quality claim
    cdir = os.path.join(Log.logdir, "steps200", "compressed")
    names = sorted(os.listdir(os.path.join(cdir, "module")))
    assert names == sorted(c["name"] for c in misc.divide_data(vol, "total_1_2_2")[0])
    top = config.load(os.path.join(cdir, "sideinfos.yaml"))
    assert top["chunks_numbers"] == 4 and list(top["data_shape"]) == list(vol.shape)
    for n in names:
        assert os.path.exists(os.path.join(cdir, "module", n, "module", "bias-0-%d" % config.load(os.path.join(cdir, "sideinfos", n, "sideinfos.yaml"))["phi_features"]))
    merged = read_img(os.path.join(Log.logdir, "steps200", "decompressed", "blk_decompressed.tif"))
    d = merged.astype(np.float64) - vol.astype(np.float64)
    assert abs(-10 * np.log10((d * d).mean() / 65535.0 ** 2) - res[200]["psnr"]) < 1e-6
    # decompress_divide (main.py:299-320): the stored artefact tree alone reproduces the merged volume bit for bit
    again = fw.decompress_divide(os.path.join(cdir, "sideinfos.yaml"), os.path.join(cdir, "module"), os.path.join(cdir, "sideinfos"))
    assert again.dtype == merged.dtype and np.array_equal(again, merged)
    # main.py:622-631: the projections of the merged volume (assembled from the ranks' z-slabs)
    mdir = os.path.join(Log.logdir, "steps200", "mip")
    for tag, v in (("blk", vol), ("blk_decompressed", merged)):
        for ax, axis in zip("dhw", (0, 1, 2)):
            for e in (".tif", ".png"):
                assert np.array_equal(np.squeeze(read_img(os.path.join(mdir, "%s_mip_%s%s" % (tag, ax, e)))), v.max(axis)[..., 0]), (tag, ax, e)


def test_dividetask_cotrained_blocks_equal_serial_blocks(tmp_path, monkeypatch):
    """the blocks a rank owns are trained together (brief_multi_fit, one HIP stream per block); every weight
    file must be byte-identical to the block-after-block run (BRIEF_COTRAIN=0)."""
    from brief_pytorch_amd.synthetic import make_volume
    vol = make_volume((16, 48, 32), seed=5)
    path = str(tmp_path / "blk.tif")
    save_img(path, vol)
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("BRIEF_COTRAIN", mode)
        opt = _opt(tmp_path / ("m" + mode), 150, "50,100", 30000.0)
        cf = opt.CompressFramework
        cf.Compress.divide.divide_type = "total_1_3_2"
        cf.Compress.divide.param_alloc = "by_size"
        Log = MyLogger(**opt.Log)
        torch.manual_seed(42)
        fw = NFGR(cf, Log=Log)
        res = fw.compress_divide(path, opt)
        assert sorted(res) == [50, 100, 150]
        files = {}
        for k in (50, 100, 150):
            mdir = os.path.join(Log.logdir, "steps%d" % k, "compressed", "module")
            for blk in sorted(os.listdir(mdir)):
                for f in sorted(os.listdir(os.path.join(mdir, blk, "module"))):
                    files[(k, blk, f)] = open(os.path.join(mdir, blk, "module", f), "rb").read()
        runs[mode] = (files, {k: res[k]["psnr"] for k in res})
        Log.close()
    assert len(runs["1"][0]) == 3 * 6 * 10 and runs["1"][0] == runs["0"][0]
    assert runs["1"][1] == runs["0"][1]


def test_dividetask_windowed_cube_blocks_do_not_depend_on_their_neighbours(tmp_path, monkeypatch):
    """Blocks with a WINDOWED cube sampler (host-drawn window numbers): every block draws from a private fork of the CPU
    generator taken after a per-block reproduc(seed) (main.py:573, 653-661: one process per block in the reference), so its
    weight files are the same whether it is co-trained, trained block after block, trained with other checkpoint spacing,
    or fitted on its own as a SingleTask of the sub-volume with the block's budget."""
    from brief_pytorch_amd.synthetic import make_volume
    vol = make_volume((16, 48, 32), seed=6)
    path = str(tmp_path / "blk.tif")
    save_img(path, vol)

    def files_of(logdir, k):
        mdir = os.path.join(logdir, "steps%d" % k, "compressed", "module")
        return {(blk, f): open(os.path.join(mdir, blk, "module", f), "rb").read() for blk in sorted(os.listdir(mdir))
                for f in sorted(os.listdir(os.path.join(mdir, blk, "module")))}

    runs = {}
    for tag, mode, ckpt in (("co", "1", "none"), ("serial", "0", "none"), ("co_ckpt", "1", "30,70")):
        monkeypatch.setenv("BRIEF_COTRAIN", mode)
        opt = _opt(tmp_path / tag, 120, ckpt, 30000.0)
        cf = opt.CompressFramework
        cf["_seed"] = 42
        cf.Compress.divide.divide_type = "total_1_3_2"
        cf.Compress.divide.param_alloc = "by_size"
        cf.Compress.sampler.cube_len = [8, 8, 8]
        cf.Compress.sampler.cube_count = 6
        Log = MyLogger(**opt.Log)
        torch.manual_seed(7)                 # whatever the caller's generator holds: the blocks reseed from the job's seed
        NFGR(cf, Log=Log).compress_divide(path, opt)
        runs[tag] = files_of(Log.logdir, 120)
        Log.close()
    assert len(runs["co"]) == 6 * 10
    assert runs["co"] == runs["serial"] == runs["co_ckpt"]
    # one block on its own: SingleTask on the sub-volume with the block's budget, after reproduc(42)
    chunks = misc.divide_data(vol, "total_1_3_2")[0]
    blk = chunks[3]
    side = config.load(os.path.join(str(tmp_path / "co" / "outputs"), os.listdir(str(tmp_path / "co" / "outputs"))[0],
                                    "steps120", "compressed", "sideinfos", blk["name"], "sideinfos.yaml"))
    opt = _opt(tmp_path / "solo", 120, "none", 0.0)
    cf = opt.CompressFramework
    cf["_seed"] = 42
    cf.Compress.sampler.cube_len = [8, 8, 8]
    cf.Compress.sampler.cube_count = 6
    cf.Compress.param.given_size = 4.0 * SIREN.calc_param_count(3, 1, side["phi_features"], cf.Module.phi.layers)
    sub_path = str(tmp_path / (blk["name"] + ".tif"))
    save_img(sub_path, np.ascontiguousarray(blk["data"]))
    Log = MyLogger(**opt.Log)
    torch.manual_seed(42)
    NFGR(cf, Log=Log).compress(sub_path)
    mdir = os.path.join(Log.logdir, "steps120", "compressed", "module")
    solo = {f: open(os.path.join(mdir, f), "rb").read() for f in sorted(os.listdir(mdir))}
    assert solo == {f: raw for (b, f), raw in runs["co"].items() if b == blk["name"]}


def test_dividetask_exception_overrides_one_block(tmp_path):
    """Compress.divide.exception (main.py:535-537, 568-569): a partial option tree merged into ONE block's task options.
    The other blocks' weight files stay byte-identical to the run without it; the overridden block is fitted with its own
    learning rate and loss and therefore differs."""
    from brief_pytorch_amd.synthetic import make_volume
    vol = make_volume((16, 32, 48), seed=9)
    path = str(tmp_path / "blk.tif")
    save_img(path, vol)
    names = [c["name"] for c in misc.divide_data(vol, "total_1_2_2")[0]]
    runs = {}
    for tag in ("plain", "exc"):
        opt = _opt(tmp_path / tag, 120, "none", 24000.0)
        cf = opt.CompressFramework
        cf.Compress.divide.divide_type = "total_1_2_2"
        cf.Compress.divide.param_alloc = "by_size"
        if tag == "exc":
            cf.Compress.divide.exception = config.to_opt({names[1]: {"CompressFramework": {"Compress": {"lr_phi": 0.004, "loss": {"name": "datasmoothl1"}}}}})
        Log = MyLogger(**opt.Log)
        torch.manual_seed(42)
        res = NFGR(cf, Log=Log).compress_divide(path, opt)
        mdir = os.path.join(Log.logdir, "steps120", "compressed", "module")
        runs[tag] = ({(blk, f): open(os.path.join(mdir, blk, "module", f), "rb").read() for blk in sorted(os.listdir(mdir))
                      for f in sorted(os.listdir(os.path.join(mdir, blk, "module")))}, res[120]["psnr"])
    plain, exc = runs["plain"][0], runs["exc"][0]
    assert sorted(plain) == sorted(exc) and len({b for b, _ in plain}) == 4
    for (blk, f), raw in plain.items():
        assert (raw == exc[(blk, f)]) == (blk != names[1]), (blk, f)
    assert runs["exc"][1] > 20
    # an override that changes the checkpoint schedule of one block cannot be stored in the job's tree: refused up front
    opt = _opt(tmp_path / "bad", 120, "none", 24000.0)
    cf = opt.CompressFramework
    cf.Compress.divide.divide_type = "total_1_2_2"
    cf.Compress.divide.exception = config.to_opt({names[0]: {"CompressFramework": {"Compress": {"max_steps": 60}}}})
    with pytest.raises(ValueError, match="exception"):
        NFGR(cf, Log=MyLogger(**opt.Log)).compress_divide(path, opt)


def test_dividetask_vessel_shaped(tmp_path):
    """BASELINE config 5 in small: opt/DivideTask/vessel.yaml (adaptotal, <= 4 blocks sized by cal_divide_num,
    by_size budget, 7-layer nets, w0 = 10) on a 16x128x128 vessel-like stack, 400 steps."""
    opt = config.load(os.path.join(ROOT, "opt", "DivideTask", "vessel.yaml"))
    from brief_pytorch_amd.synthetic import ensure_dataset
    path = ensure_dataset(str(tmp_path / "dataset" / "synthetic_vessel_16x128x128.tif"))
    vol = read_img(path)
    assert vol.shape == (16, 128, 128, 1) and 0.02 < (vol > 4000).mean() < 0.12      # sparse foreground
    cf = opt.CompressFramework
    cf.Compress.max_steps = 400
    cf.Compress.checkpoints = "none"
    cf.Compress.param.filesize_ratio = 32
    opt.Log.outputs_dir = str(tmp_path / "outputs")
    opt.Log.time = False
    Log = MyLogger(**opt.Log)
    torch.manual_seed(42)
    fw = NFGR(cf, Log=Log)
    res = fw.compress_divide(path, opt)
    cdir = os.path.join(Log.logdir, "steps400", "compressed")
    names = sorted(os.listdir(os.path.join(cdir, "module")))
    assert 1 <= len(names) <= 4
    feats = [config.load(os.path.join(cdir, "sideinfos", n, "sideinfos.yaml"))["phi_features"] for n in names]
    total = sum(SIREN.calc_param_count(3, 1, f, 7) for f in feats) * 4
    assert abs(total - vol.size * 2 / 32) / (vol.size * 2 / 32) < 0.08            # the blocks share the ratio-32 budget
    assert res[400]["psnr"] > 27                                                   # sanity floor after 400 steps (29.7 measured)
    merged = read_img(os.path.join(Log.logdir, "steps400", "decompressed", "synthetic_vessel_16x128x128_decompressed.tif"))
    d = merged.astype(np.float64) - vol.astype(np.float64)
    assert abs(-10 * np.log10((d * d).mean() / 65535.0 ** 2) - res[400]["psnr"]) < 1e-6


def test_cli_singletask_end_to_end(tmp_path, monkeypatch):
    """python main.py -p <yaml>: the drop-in CLI on a generated 24x32x40 volume (randomcube -> full batch)"""
    import subprocess
    import sys
    opt = config.load(os.path.join(ROOT, "opt", "SingleTask", "default.yaml"))
    opt.Dataset.data_path = str(tmp_path / "dataset" / "synthetic_24x32x40.tif")
    opt.CompressFramework.Compress.max_steps = 150
    opt.CompressFramework.Compress.checkpoints = "every_50"
    opt.Log.outputs_dir = str(tmp_path / "outputs")
    opt.Log.time = False
    y = str(tmp_path / "cli.yaml")
    config.save(opt, y)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "-p", y, "-g", "0"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "steps 150" in r.stdout and "psnr" in r.stdout
    run = os.path.join(str(tmp_path / "outputs"), "single")
    # main.py:452-453, 695: without -stepstore only the last step directory survives (all three were evaluated: performance.csv)
    assert sorted(d for d in os.listdir(run) if d.startswith("steps")) == ["steps150"]
    assert len(open(os.path.join(run, "performance.csv")).read().strip().splitlines()) == 4
    opt.Log.project_name = "kept"
    config.save(opt, y)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "-p", y, "-g", "0", "-stepstore"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert sorted(d for d in os.listdir(os.path.join(str(tmp_path / "outputs"), "kept")) if d.startswith("steps")) == ["steps100", "steps150", "steps50"]
    assert os.path.exists(os.path.join(run, "steps150", "compressed", "module", "weight-0-21-3")) or \
        any(f.startswith("weight-0-") for f in os.listdir(os.path.join(run, "steps150", "compressed", "module")))
    assert os.path.exists(os.path.join(run, "performance.csv"))


def test_cli_dividetask_two_ranks(tmp_path):
    """DivideTask under torch.distributed with 2 ranks (sharing this box's GPU, gloo for the reductions):
    blocks are split over the ranks, PSNR comes from the all-reduced [SSE, n], rank 0 merges and writes."""
    import subprocess
    import sys
    opt = config.load(os.path.join(ROOT, "opt", "DivideTask", "default.yaml"))
    opt.Dataset.data_path = str(tmp_path / "dataset" / "synthetic_16x48x64.tif")
    cf = opt.CompressFramework
    cf.Compress.divide.divide_type = "total_1_2_2"
    cf.Compress.divide.param_alloc = "by_size"
    cf.Compress.max_steps = 120
    cf.Compress.checkpoints = "none"
    cf.Compress.param.filesize_ratio = 0
    cf.Compress.param.given_size = 24000
    cf.Module.phi.layers = 4
    opt.Log.outputs_dir = str(tmp_path / "outputs")
    y = str(tmp_path / "div.yaml")
    config.save(opt, y)
    env = dict(os.environ, BRIEF_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "main.py"), "-p", y]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    if r.returncode != 0:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", "divide2_stderr.log"), "w").write(r.stdout + "\n----\n" + r.stderr)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "steps 120" in r.stdout and "psnr" in r.stdout
    runs = sorted(os.listdir(str(tmp_path / "outputs")))
    assert len(runs) == 1 and runs[0].startswith("divide_")       # ONE (timestamped) run directory, chosen by rank 0
    run = os.path.join(str(tmp_path / "outputs"), runs[0])
    cdir = os.path.join(run, "steps120", "compressed")
    assert len(os.listdir(os.path.join(cdir, "module"))) == 4
    merged = read_img(os.path.join(run, "steps120", "decompressed", "synthetic_16x48x64_decompressed.tif"))
    vol = read_img(opt.Dataset.data_path)
    d = merged.astype(np.float64) - vol.astype(np.float64)
    psnr = -10 * np.log10((d * d).mean() / 65535.0 ** 2)
    import csv
    rows = list(csv.DictReader(open(os.path.join(run, "performance.csv"))))
    assert abs(float(rows[-1]["psnr"]) - psnr) < 1e-6 and psnr > 20


def _run_divide(tmp_path, vol, divide_type, steps=120, given=60000.0, sub="d", mutate=None, ckpt="none"):
    path = str(tmp_path / (sub + ".tif"))
    save_img(path, vol)
    opt = _opt(tmp_path / sub, steps, ckpt, given)
    cf = opt.CompressFramework
    cf.Compress.divide.divide_type = divide_type
    cf.Compress.divide.param_alloc = "by_size"
    if mutate is not None:
        mutate(cf)
    Log = MyLogger(**opt.Log)
    torch.manual_seed(42)
    fw = NFGR(cf, Log=Log)
    res = fw.compress_divide(path, opt)
    return fw, Log, res, path


def test_dividetask_adaptive_octree_mixed_blocks_and_pruned_region(tmp_path):
    """BASELINE configs 4 / 5 in small, through the real DivideTask path: divide_type adaptive_* with Nb >= 8 (the octree +
    tree-knapsack route, main.py:456-507), blocks of different sizes, an all-zero region that is pruned (never fitted) and
    decodes as zeros (utils/adaptive_blocking.py:341-352, utils/misc.py:432), z-sharded decode from the stored artefacts"""
    from brief_pytorch_amd.synthetic import make_volume
    vol = make_volume((32, 64, 64), seed=52)
    vol[16:32, 32:64, 32:64] = 0                       # the golden case "c" of tests/golden/adaptive.npz
    fw, Log, res, path = _run_divide(tmp_path, vol, "adaptive_-1_-1_0_0_20", steps=150, given=120000.0)
    cdir = os.path.join(Log.logdir, "steps150", "compressed")
    names = sorted(os.listdir(os.path.join(cdir, "module")))
    sizes = set()
    cover = np.zeros(vol.shape[:3], np.int32)
    for n in names:
        r = misc.parse_chunk_name(n)
        sizes.add((r["d"][1] - r["d"][0] + 1, r["h"][1] - r["h"][0] + 1, r["w"][1] - r["w"][0] + 1))
        cover[r["d"][0]:r["d"][1] + 1, r["h"][0]:r["h"][1] + 1, r["w"][0]:r["w"][1] + 1] += 1
    assert 8 <= len(names) <= 20 and len(sizes) >= 2                   # mixed block sizes
    assert cover[16:32, 32:64, 32:64].max() == 0 and cover.max() == 1   # the zero region belongs to no block
    assert (cover.sum() + 16 * 32 * 32) == vol[..., 0].size
    side = config.load(os.path.join(cdir, "sideinfos.yaml"))
    assert side["chunks_numbers"] == len(names) and list(side["data_shape"]) == list(vol.shape)
    merged = read_img(os.path.join(Log.logdir, "steps150", "decompressed", "d_decompressed.tif"))
    assert (merged[16:32, 32:64, 32:64] == 0).all()
    d = merged.astype(np.float64) - vol.astype(np.float64)
    assert abs(-10 * np.log10((d * d).mean() / 65535.0 ** 2) - res[150]["psnr"]) < 1e-6
    from oracle import oracle as O
    assert abs(O.ssim(vol.astype(np.float32), merged.astype(np.float32), 65535) - res[150]["ssim"]) < 5e-5
    again = fw.decompress_divide(os.path.join(cdir, "sideinfos.yaml"), os.path.join(cdir, "module"), os.path.join(cdir, "sideinfos"))
    assert np.array_equal(again, merged)
    assert res[150]["psnr"] > 25


def test_metrics_are_taken_against_the_original_when_denoise_changes_the_data(tmp_path):
    """preprocess.denoise.level > 0 zeroes dim voxels before the fit; PSNR / the MIP images compare the decode with the
    UNTOUCHED file (the reference re-reads it, main.py:433, 624), and the caller's array is not written to"""
    from brief_pytorch_amd.synthetic import make_volume
    vol = make_volume((12, 20, 24), seed=8)
    level = int(np.quantile(vol, 0.3))
    keep = vol.copy()

    def mut(cf):
        cf.Compress.preprocess.denoise.level = level
        cf.Compress.preprocess.denoise.close = False       # plain threshold (with close = [2,2,2] the opening erases this noise-like mask)
    fw, Log, res, path = _run_divide(tmp_path, vol, "total_1_2_2", steps=100, given=30000.0, mutate=mut)
    assert np.array_equal(vol, keep)
    merged = read_img(os.path.join(Log.logdir, "steps100", "decompressed", "d_decompressed.tif"))
    d = merged.astype(np.float64) - keep.astype(np.float64)
    assert abs(-10 * np.log10((d * d).mean() / 65535.0 ** 2) - res[100]["psnr"]) < 1e-6
    # SingleTask: same rule
    opt = _opt(tmp_path / "s", 80, "none", 8000.0)
    opt.CompressFramework.Compress.preprocess.denoise.level = level
    opt.CompressFramework.Compress.preprocess.denoise.close = False
    Log2 = MyLogger(**opt.Log)
    torch.manual_seed(1)
    r2 = NFGR(opt.CompressFramework, Log=Log2).compress(path)
    dec = read_img(os.path.join(Log2.logdir, "steps80", "decompressed", "d_decompressed.tif"))
    d = dec.astype(np.float64) - keep.astype(np.float64)
    assert abs(-10 * np.log10((d * d).mean() / 65535.0 ** 2) - r2[80]["psnr"]) < 1e-4
    pre = read_img(os.path.join(Log2.logdir, "d_preprocessed.tif"))
    assert (pre[keep <= level] == 0).all() and np.array_equal(pre[keep > level], keep[keep > level]) and (pre == 0).mean() > 0.25


def test_dividetask_steplr_and_exp_weights_and_cycliclr(tmp_path):
    """StepLR / CyclicLR blocks cannot ride brief_fit_job (MultiStepLR only): the DivideTask falls back to the per-block
    loop instead of failing; an 'exp_x_v' loss-weight map (float64 out of numpy) reaches the kernel as float32"""
    from brief_pytorch_amd.synthetic import make_volume
    vol = make_volume((12, 24, 24), seed=9)

    def mut(cf):
        cf.Compress.lr_scheduler_phi = config.to_opt({"name": "StepLR", "step_size": 30, "gamma": 0.5})
        cf.Compress.loss.weight = ["exp_20000_0.5"]
        cf.Compress.loss.weight_thres = 0
    fw, Log, res, _ = _run_divide(tmp_path, vol, "total_1_2_2", steps=90, given=30000.0, mutate=mut)
    assert res[90]["psnr"] > 20

    def mut2(cf):
        cf.Compress.lr_scheduler_phi = config.to_opt({"name": "CyclicLR", "base_lr": 2e-4, "max_lr": 2e-3, "step_size_up": 20})
    fw, Log, res2, _ = _run_divide(tmp_path, vol, "total_1_2_2", steps=90, given=30000.0, sub="c", mutate=mut2)
    assert res2[90]["psnr"] > 20
    w = misc.parse_weight(vol, ["exp_20000_0.5"])
    assert w.dtype == np.float64                       # what numpy hands back: the framework must cast
    from brief_pytorch_amd import _lib
    from brief_pytorch_amd.fit import Fitter
    m = SIREN(features=16, layers=3).to("cuda")
    tv = torch.rand(vol.size, 1, device="cuda")
    with pytest.raises(_lib.BriefError):
        Fitter(m, tv, vol.shape[:3], weights=torch.from_numpy(w.reshape(-1, 1)).cuda())
    with pytest.raises(_lib.BriefError):
        m.train_step(100, tv.double(), grid=(vol.shape[:3], -1.0, 1.0))


def test_init_net_path_warm_start_and_half_option(tmp_path):
    """param.init_net_path (main.py:349-354: weights only, no optimizer state) and Compress.half (mapped to the bf16
    matrix pipe with the reference's 2-bytes-per-parameter budget rule, main.py:217, 242)"""
    from brief_pytorch_amd.synthetic import make_volume
    vol = make_volume((16, 24, 32), seed=10)
    path = str(tmp_path / "w.tif")
    save_img(path, vol)
    given = 4.0 * SIREN.calc_param_count(3, 1, 24, 4)
    opt = _opt(tmp_path / "a", 200, "none", given)
    opt.CompressFramework.Module.phi.layers = 4
    Log = MyLogger(**opt.Log)
    torch.manual_seed(42)
    r1 = NFGR(opt.CompressFramework, Log=Log).compress(path)
    mod = os.path.join(Log.logdir, "steps200", "compressed", "module")
    # warm start from those weights: step 0 of the second run starts where the first ended
    opt2 = _opt(tmp_path / "b", 1, "none", given)
    opt2.CompressFramework.Module.phi.layers = 4
    opt2.CompressFramework.Compress.param.init_net_path = mod
    opt2.CompressFramework.Compress.lr_phi = 0.0
    Log2 = MyLogger(**opt2.Log)
    torch.manual_seed(7)                                   # a different seed: the init must come from the files
    r2 = NFGR(opt2.CompressFramework, Log=Log2).compress(path)
    assert abs(r2[1]["psnr"] - r1[200]["psnr"]) < 1e-9 and abs(r2[1]["ssim"] - r1[200]["ssim"]) < 1e-12
    a = read_img(os.path.join(Log.logdir, "steps200", "decompressed", "w_decompressed.tif"))
    b = read_img(os.path.join(Log2.logdir, "steps1", "decompressed", "w_decompressed.tif"))
    assert np.array_equal(a, b)
    # Compress.half: bf16 path, 2 bytes per parameter in the budget -> a wider net for the same bytes
    opt3 = _opt(tmp_path / "h", 150, "none", given)
    opt3.CompressFramework.Module.phi.layers = 4
    opt3.CompressFramework.Compress.half = True
    Log3 = MyLogger(**opt3.Log)
    torch.manual_seed(42)
    fw3 = NFGR(opt3.CompressFramework, Log=Log3)
    assert fw3.precision == "bf16"
    r3 = fw3.compress(path)
    side = config.load(os.path.join(Log3.logdir, "steps150", "compressed", "sideinfos.yaml"))
    assert side["phi_features"] == SIREN.calc_features(given / 2.0, 3, 1, 4) > 24 and side["phi_precision"] == "bf16"
    assert r3[150]["psnr"] > 25


def test_bench_dividetask_path_on_the_rccl_backend_world_size_one():
    """the code a multi-GPU SCALE run executes, on the one GPU of the test box: bench.py --divide initialises the **nccl**
    (RCCL) process group with one rank and drives NFGR.compress_divide through divide_bench: broadcast_object_list of the
    block list and the run directory, device-tensor all-reduces ([SSE, SSIM-sum, slices, voxels], the timed-window MAX),
    barriers with device_id set.  Also: --gpus N must equal the launcher's world size, and --gpus N without a launcher
    starts the ranks itself (refused here: one device)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--divide", "--block", "128", "--steps", "4", "--warmup", "2", "--preroll", "4",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["metric"] == "encode_voxels_per_sec" and out["value"] > 1e6 and out["scaling"] == "weak"
    assert out["config"]["volume"] == [128, 128, 128] and "DivideTask" in out["config"]["workload"]
    assert 10.0 < out["psnr_at_bitrate"]["psnr_db"] < 100.0 and 0.0 < out["psnr_at_bitrate"]["ssim"] <= 1.0
    assert 0.0 < out["roofline"]["frac"] < 1.0
    # every rank's identity travels in the line: rank, the group's world size and backend, the device (UUID, index, CU count) — what lets a
    # SCALE record prove N ranks on N devices
    assert len(out["ranks"]) == 1 and out["ranks"][0]["backend"] == "nccl" and out["ranks"][0]["world_size"] == 1 and out["distinct_devices"] == 1
    assert out["ranks"][0]["device"]["compute_units"] >= 1 and out["ranks"][0]["device"]["index"] == 0
    # a launcher whose world size differs from --gpus is an error, not a silent one-rank run
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300,
                        env={**env, "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r2.returncode == 2 and "WORLD_SIZE" in r2.stderr
    if torch.cuda.device_count() < 2:
        r3 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300, env=env)
        assert r3.returncode == 2 and "device" in r3.stderr
    # the self-launch itself, rehearsed with two gloo ranks sharing this GPU: bench.py starts `torch.distributed.run` as a child,
    # the two ranks run the DivideTask path on a 2 x 128^3 volume and rank 0 prints the one line
    r4 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--block", "128", "--steps", "4", "--warmup", "2", "--preroll", "4",
                         "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env={**env, "BRIEF_DIST_BACKEND": "gloo", "BRIEF_SHARE_GPU": "1"})
    assert r4.returncode == 0, r4.stderr[-3000:]
    out2 = json.loads([l for l in r4.stdout.splitlines() if l.startswith("{")][-1])
    assert out2["n_gpus"] == 2 and out2["config"]["volume"] == [256, 128, 128] and out2["value"] > 1e6 and 10.0 < out2["psnr_at_bitrate"]["psnr_db"] < 100.0
    assert [r["rank"] for r in out2["ranks"]] == [0, 1] and all(r["world_size"] == 2 and r["backend"] == "gloo" for r in out2["ranks"])
    assert len({r["pid"] for r in out2["ranks"]}) == 2 and out2["distinct_devices"] == 1      # (the rehearsal shares the one device; under nccl more ranks than devices are refused)


def test_bench_four_ranks_share_the_gpu_like_a_scale_run():
    """the SCALE run's plumbing at the largest rank count this pool lets one box rehearse: at most 6 processes may hold the card at
    once and the test runner itself is one of them, so four ranks (the 8-rank partition / LPT / all-reduce pattern itself runs on the
    CPU in tests/test_dist_cpu.py).  bench.py --gpus 4 starts four gloo ranks that share the GPU, rank 0 partitions a 4 x 64^3 volume
    and broadcasts the block list, every rank writes its slab of the shared file, fits the block LPT gives it, evaluates its z-slab,
    and the [SSE, SSIM-sum, slices, voxels] all-reduce feeds the one JSON line with n_gpus = 4."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--block", "64", "--steps", "3", "--warmup", "1", "--preroll", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env={**env, "BRIEF_DIST_BACKEND": "gloo", "BRIEF_SHARE_GPU": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 4 and out["config"]["volume"] == [256, 64, 64] and out["scaling"] == "weak" and out["value"] > 1e5
    assert 10.0 < out["psnr_at_bitrate"]["psnr_db"] < 100.0 and 0.0 < out["psnr_at_bitrate"]["ssim"] <= 1.0


def test_adaptive_octree_device_statistics_choose_the_pinned_partition():
    """volumes of 2^24 voxels or more take their octree variances and FFT features from the GPU (torch float64 reductions,
    rocFFT) instead of numpy, whose arithmetic tests/golden/adaptive.npz pins: on a 128^3 volume with an all-zero octant
    both routes must prune the same nodes, agree on every node's feature to 1e-9 relative, and choose the SAME partition
    (the knapsack only sees differences that flip a comparison)."""
    from brief_pytorch_amd import adaptive_blocking as ab
    from brief_pytorch_amd.synthetic import make_volume
    vol = make_volume((128, 128, 128), seed=61)
    vol[64:, :64, 64:] = 0
    data = vol
    Nb, minl, maxl = ab.adaptive_levels(20, 0, 3)
    trees = {}
    for dev in (None, "cuda"):
        root = ab.build_tree(data.shape, maxl, 3)
        ab.prune_and_score(root, data, 0, 0, dim=3, device=dev)
        active, best = ab.solve_tree(root, Nb, minl, 3)
        trees[dev] = (root, sorted((p.z, p.y, p.x, p.d, p.h, p.w) for p in active), best)
    nodes_a, nodes_b = list(ab.iter_nodes(trees[None][0])), list(ab.iter_nodes(trees["cuda"][0]))
    assert len(nodes_a) == len(nodes_b) and [n.pruned for n in nodes_a] == [n.pruned for n in nodes_b]
    assert any(n.pruned for n in nodes_a)
    fa = np.array([n.feature for n in nodes_a if not n.pruned]); fb = np.array([n.feature for n in nodes_b if not n.pruned])
    assert np.max(np.abs(fa - fb) / np.abs(fa)) < 1e-9
    assert trees[None][1] == trees["cuda"][1] and abs(trees[None][2] - trees["cuda"][2]) <= 1e-9 * abs(trees[None][2])


def test_device_normalisation_is_bit_identical_to_the_host_rule():
    """io.normalize_data_device (what NFGR.prepare_fit uses) against io.normalize_data (utils/io.py:65-80, pinned to the
    reference's goldens by tests/test_oracle_golden.py): same bits, same side information, for uint16 / uint8 / float32 data
    and for a rule that keeps the host path"""
    from brief_pytorch_amd.io import normalize_data, normalize_data_device
    from brief_pytorch_amd.misc import weight_is_unit
    from brief_pytorch_amd.synthetic import make_volume
    rng = np.random.default_rng(3)
    cases = [("minmaxany_0_100", make_volume((24, 40, 56), seed=4)),
             ("minmaxany_-1_1", (rng.integers(3, 250, size=(33, 47, 3))).astype(np.uint8)),
             ("minmaxany_0_100", rng.normal(5.0, 3.0, size=(8, 16, 16, 1)).astype(np.float32)),
             ("minmax01_0mean", make_volume((8, 16, 16), seed=5))]
    for name, data in cases:
        host, side_h = normalize_data(data, name)
        dev, side_d = normalize_data_device(data, name, "cuda")
        assert dev.is_cuda and dev.dtype == torch.float32 and torch.equal(dev.cpu(), host), name
        assert side_h.keys() == side_d.keys() and all(side_h[k] == side_d[k] for k in side_h), (name, side_h, side_d)
    assert weight_is_unit(["value_65535_65535_1"]) and weight_is_unit(["none"]) and weight_is_unit([])
    assert not weight_is_unit(["value_0_100_2"]) and not weight_is_unit(["exp_100_0.5"]) and not weight_is_unit(["quantile_0_0.1_0.9_3"])


def test_optional_precision_falls_back_to_fp32_before_any_work_when_the_net_is_too_wide(tmp_path, caplog):
    """Compress.precision bf16x3 has kernels up to 256 features, bf16 up to 512.  A budget that solves to a wider net (in a
    DivideTask: one large block) must not abort the job after the partition: the net is built in fp32, a warning says so, and the
    artefact records the precision THIS net was fitted in, so that the decoder evaluates it the same way."""
    import logging
    from brief_pytorch_amd.synthetic import make_volume
    vol = make_volume((12, 24, 24), seed=12)
    path = str(tmp_path / "v.tif")
    save_img(path, vol)
    opt = _opt(tmp_path, 40, "none", 4.0 * SIREN.calc_param_count(3, 1, 288, 3))
    cf = opt.CompressFramework
    cf.Module.phi.layers = 3
    cf.Compress.precision = "bf16x3"
    Log = MyLogger(**opt.Log)
    torch.manual_seed(42)
    fw = NFGR(cf, Log=Log)
    with caplog.at_level(logging.WARNING):
        res = fw.compress(path)
    assert fw.module["phi"].features == 288 and fw.module["phi"].precision == "fp32" and fw.module_precision == "fp32"
    assert any("runs in fp32" in r.getMessage() for r in caplog.records)
    side = config.load(os.path.join(Log.logdir, "steps40", "compressed", "sideinfos.yaml"))
    assert side["phi_features"] == 288 and side["phi_precision"] == "fp32"
    dec = NFGR.decompress(config.to_opt({"CompressFramework": cf}), os.path.join(Log.logdir, "steps40", "compressed", "module"), dict(side))
    assert dec.shape == vol.shape and np.isfinite(res[40]["psnr"])
