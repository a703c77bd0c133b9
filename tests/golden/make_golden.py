#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

Runs only in the build container (needs /root/reference, CPU torch).  Nothing of
the reference is copied: this script stubs the third-party packages the
reference imports but never uses on the SIREN fit/decode path (SURVEY.md
Appendix A), imports the reference modules in place, drives them on small seeded
inputs and stores inputs + outputs as .npz fixtures.  The GPU box never runs
this file; tests only read the .npz files.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
"""
import copy
import os
import sys
import types

sys.dont_write_bytecode = True
REF = os.environ.get("BRIEF_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))

if not os.path.isdir(REF):
    sys.exit("make_golden.py: reference tree %s not present; goldens can only be regenerated in the build container" % REF)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Dummy:
    def __init__(self, *a, **k):
        pass


# third-party packages absent from the image and unused by the SIREN path
_stub("compressai")
_stub("compressai.entropy_models", EntropyBottleneck=_Dummy, GaussianConditional=_Dummy)
_stub("cv2")
_stub("gurobipy")
_stub("tifffile")
_stub("pynvml", nvmlInit=lambda: None)
_stub("prettytable", PrettyTable=_Dummy)
_stub("py7zr", FILTER_BZIP2=1, FILTER_LZMA=2, FILTER_ZSTD=3)
_oc = _stub("omegaconf", OmegaConf=_Dummy)
_oc.listconfig = _stub("omegaconf.listconfig", ListConfig=list)
_oc.dictconfig = _stub("omegaconf.dictconfig", DictConfig=dict)
_stub("skimage")
_stub("skimage.metrics", structural_similarity=None)
import torch  # noqa: E402
import torch.utils  # noqa: E402
_tb = _stub("torch.utils.tensorboard", SummaryWriter=_Dummy)
torch.utils.tensorboard = _tb

sys.path.insert(0, REF)
sys.path.insert(1, REPO)
import numpy as np  # noqa: E402
import yaml  # noqa: E402

from utils.Networks import SIREN, init_phi  # noqa: E402
import main as refmain  # noqa: E402
from utils import misc as refmisc  # noqa: E402
from utils import io as refio  # noqa: E402
from utils import dataset as refdataset  # noqa: E402
from utils.adaptive_blocking import cal_divide_num, cal_feature  # noqa: E402
from brief_pytorch_amd.synthetic import make_volume  # noqa: E402  (this repo's generator)

torch.set_num_threads(int(os.environ.get("BRIEF_GOLDEN_THREADS", "1")))  # 1 = fixed reduction order (the long half/f32 fits may use more)


class AttrDict(dict):
    """Minimal OmegaConf stand-in: attribute access, AttributeError on a miss."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_attr(o):
    if isinstance(o, dict):
        return AttrDict({k: to_attr(v) for k, v in o.items()})
    if isinstance(o, list):
        return [to_attr(v) for v in o]
    return o


def load_opt():
    with open(os.path.join(REF, "opt/SingleTask/default.yaml")) as f:
        return to_attr(yaml.safe_load(f))


def state_arrays(model):
    """weights/biases in layer order as numpy, the ModelSave.py file contract."""
    out = {}
    for l in range(len(model.net)):
        out["w%d" % l] = model.net[l][0].weight.detach().numpy().copy()
        out["b%d" % l] = model.net[l][0].bias.detach().numpy().copy()
    return out


def grads_arrays(model):
    out = {}
    for l in range(len(model.net)):
        out["gw%d" % l] = model.net[l][0].weight.grad.detach().numpy().copy()
        out["gb%d" % l] = model.net[l][0].bias.grad.detach().numpy().copy()
    return out


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def build(seed, layers, features, w0, cin=3, cout=1, output_act=False):
    torch.manual_seed(seed)
    return SIREN(coords_channel=cin, data_channel=cout, features=features, layers=layers, w0=w0,
                 output_act=output_act)


# ----------------------------------------------------------------------------- 1. init
def g_init():
    arrs = {}
    for seed in (0, 42):
        for (L, F) in ((3, 64), (5, 22), (7, 56)):
            m = build(seed, L, F, 20)
            for k, v in state_arrays(m).items():
                arrs["s%d_L%d_F%d_%s" % (seed, L, F, k)] = v
        m = build(seed, 5, 256, 20)
        sa = state_arrays(m)
        # (5,256) is large: keep per-tensor float64 sums + a few probes instead of the tensors
        arrs["s%d_L5_F256_sums" % seed] = np.array([sa[k].astype(np.float64).sum() for k in sorted(sa)])
        arrs["s%d_L5_F256_abs" % seed] = np.array([np.abs(sa[k]).astype(np.float64).sum() for k in sorted(sa)])
        arrs["s%d_L5_F256_w1_row0" % seed] = sa["w1"][0].copy()
        arrs["s%d_L5_F256_b3" % seed] = sa["b3"].copy()
    save("init", **arrs)


# ----------------------------------------------------------------------------- 2. forward
def probe_coords(n, cin, seed):
    g = np.random.default_rng(seed)
    c = g.uniform(-1, 1, size=(n, cin)).astype(np.float32)
    # corners first
    k = 0
    for bits in range(2 ** cin):
        c[k] = [1.0 if (bits >> j) & 1 else -1.0 for j in range(cin)]
        k += 1
    c[k] = 0.0
    return c


def g_forward():
    arrs = {}
    for tag, (L, F, w0, cin, cout, oa) in {
        "a": (3, 64, 20, 3, 1, False),
        "b": (5, 22, 20, 3, 1, False),
        "c": (7, 56, 10, 3, 1, False),
        "d": (5, 256, 20, 3, 1, False),
        "e": (4, 24, 20, 2, 3, False),   # 2-D RGB image net
        "f": (3, 32, 20, 3, 1, True),    # output_act
    }.items():
        m = build(7, L, F, w0, cin, cout, oa)
        x = probe_coords(257, cin, 11)
        with torch.no_grad():
            y = m(torch.from_numpy(x)).numpy()
        arrs[tag + "_cfg"] = np.array([L, F, w0, cin, cout, int(oa)])
        arrs[tag + "_x"] = x
        arrs[tag + "_y"] = y
        if F <= 64:
            for k, v in state_arrays(m).items():
                arrs[tag + "_" + k] = v
        else:
            arrs[tag + "_seed"] = np.array([7])
    save("forward", **arrs)


# ----------------------------------------------------------------------------- 3. loss + grads
def g_grads():
    opt = load_opt()
    arrs = {}
    N = 4096
    g = np.random.default_rng(5)
    for tag, (L, F, w0, loss_name, weighted, thr) in {
        "mse_unit": (5, 22, 20, "datal2", False, 0.0),
        "mse_unit_thr": (5, 22, 20, "datal2", False, 150.0),       # default-yaml behaviour: thr > max => all weights 1
        "mse_w_thr": (3, 64, 20, "datal2", True, 40.0),           # neuron.yaml style weights + finite thr
        "sl1_w": (7, 56, 10, "datasmoothl1", True, 0.0),
        "mse_256": (5, 256, 20, "datal2", False, 0.0),
    }.items():
        cf = copy.deepcopy(opt.CompressFramework)
        cf.Compress.loss.name = loss_name
        cf.Compress.loss.beta = 0.01
        nf = refmain.NFGR(cf)
        m = build(3, L, F, w0)
        x = g.uniform(-1, 1, size=(N, 3)).astype(np.float32)
        y = g.uniform(0, 100, size=(N, 1)).astype(np.float32)
        if weighted:
            w = np.where(g.uniform(size=(N, 1)) < 0.3, 0.1, 1.0).astype(np.float32)
        else:
            w = np.ones((N, 1), np.float32)
        wt = torch.from_numpy(w.copy())
        yhat = m(torch.from_numpy(x))
        loss = nf.loss_func(torch.from_numpy(y), yhat, wt, thr)
        loss.backward()
        arrs[tag + "_cfg"] = np.array([L, F, w0, {"datal2": 0, "datasmoothl1": 1}[loss_name], thr, 0.01])
        arrs[tag + "_x"], arrs[tag + "_y"], arrs[tag + "_w"] = x, y, w
        arrs[tag + "_yhat"] = yhat.detach().numpy()
        arrs[tag + "_loss"] = np.array([loss.item()], np.float64)
        arrs[tag + "_w_after"] = wt.numpy()
        if F <= 64:
            for k, v in state_arrays(m).items():
                arrs[tag + "_" + k] = v
            for k, v in grads_arrays(m).items():
                arrs[tag + "_" + k] = v
        else:
            ga = grads_arrays(m)
            arrs[tag + "_gsum"] = np.array([ga[k].astype(np.float64).sum() for k in sorted(ga)])
            arrs[tag + "_gabs"] = np.array([np.abs(ga[k]).astype(np.float64).sum() for k in sorted(ga)])
            arrs[tag + "_gw2_row5"] = ga["gw2"][5].copy()
            arrs[tag + "_gb1"] = ga["gb1"].copy()
            arrs[tag + "_gw0"] = ga["gw0"].copy()
            arrs[tag + "_gw4"] = ga["gw4"].copy()
    save("grads", **arrs)


# ----------------------------------------------------------------------------- 4. optimisers
def g_optim():
    arrs = {}
    N = 512
    g = np.random.default_rng(9)
    x = torch.from_numpy(g.uniform(-1, 1, size=(N, 3)).astype(np.float32))
    y = torch.from_numpy(g.uniform(0, 100, size=(N, 1)).astype(np.float32))
    arrs["x"], arrs["y"] = x.numpy(), y.numpy()
    for name in ("Adamax", "Adam", "SGD"):
        m = build(1, 4, 24, 20)
        for k, v in state_arrays(m).items():
            arrs["init_" + k] = v
        optim = refmisc.configure_optimizer(m.parameters(), name, 1e-3)
        sched = refmisc.configure_lr_scheduler(optim, AttrDict(name="MultiStepLR", milestones=[3, 6], gamma=0.2))
        for step in range(1, 11):
            optim.zero_grad()
            loss = torch.nn.functional.mse_loss(m(x), y)
            loss.backward()
            optim.step()
            sched.step()
            if step in (1, 2, 10):
                for k, v in state_arrays(m).items():
                    arrs["%s_t%d_%s" % (name, step, k)] = v
                if name != "SGD":
                    keys = ("exp_avg", "exp_inf") if name == "Adamax" else ("exp_avg", "exp_avg_sq")
                    for pi, p in enumerate(m.parameters()):
                        st = optim.state[p]
                        arrs["%s_t%d_p%d_s1" % (name, step, pi)] = st[keys[0]].numpy().copy()
                        arrs["%s_t%d_p%d_s2" % (name, step, pi)] = st[keys[1]].numpy().copy()
            arrs["%s_loss_t%d" % (name, step)] = np.array([loss.item()])
    save("optim", **arrs)


# ----------------------------------------------------------------------------- 5. fit traces
def _fit_trace(vol, layers, features, w0, sampler_name, steps, sample_size, seed, rec_idx):
    """Drive the reference loop body (main.py:385-400) and record what happened."""
    opt = load_opt()
    cf = opt.CompressFramework
    cf.Compress.gpu = False
    cf.Decompress.gpu = False
    cf.Module.phi.layers = layers
    cf.Module.phi.w0 = w0
    cf.Compress.sampler.name = sampler_name
    cf.Compress.sampler.sample_size = sample_size
    refmain.reproduc(opt.Reproduc if seed == 42 else AttrDict(seed=seed, benchmark=False, deterministic=True))
    nf = refmain.NFGR(cf)
    nf.device = "cpu"
    weight = refmisc.parse_weight(vol, cf.Compress.loss.weight)
    data, sideinfos = refio.normalize_data(vol, **cf.Normalize)
    pcount = SIREN.calc_param_count(3, 1, features, layers, False)
    phi_features, _ = nf.prepare_module(4.0 * pcount)
    assert phi_features == features, (phi_features, features)
    init = state_arrays(nf.module["phi"])
    if sampler_name == "randompoint":
        sampler = refmain.RandompointSampler(data, weight, cf.Compress.coords_mode, sample_size, steps, "cpu")
    else:
        sampler = refmain.RandomCubeSampler(data, weight, cf.Compress.coords_mode, cf.Compress.sampler.cube_count,
                                            copy.deepcopy(cf.Compress.sampler.cube_len), steps, "cpu", True)
    optim = refmisc.configure_optimizer(nf.module["phi"].parameters(), cf.Compress.optimizer_name_phi, cf.Compress.lr_phi)
    sched = refmisc.configure_lr_scheduler(optim, cf.Compress.lr_scheduler_phi)
    thr, _ = refio.normalize_data(np.array(cf.Compress.loss.weight_thres), **cf.Normalize, max=sideinfos["max"], min=sideinfos["min"])
    thr = float(thr)
    losses, idxs = [], []
    if rec_idx:
        # replay torch.randint exactly as the sampler will draw it: record by wrapping
        orig_randint = torch.randint

        def rec(*a, **k):
            r = orig_randint(*a, **k)
            idxs.append(r.numpy().copy())
            return r
        torch.randint = rec
    try:
        for c, d, w in sampler:
            optim.zero_grad()
            yhat = nf.module["phi"].forward(c)
            loss = nf.loss_func(d, yhat, w, thr)
            loss.backward()
            optim.step()
            sched.step()
            losses.append(loss.item())
    finally:
        if rec_idx:
            torch.randint = orig_randint
    final = state_arrays(nf.module["phi"])
    return nf, cf, sideinfos, init, final, np.array(losses, np.float64), (np.stack(idxs) if idxs else None), thr


def g_trace():
    arrs = {}
    # (i) full batch 16^3 via randomcube (one cube == the volume)
    vol = make_volume((16, 16, 16), seed=42)
    nf, cf, side, init, final, losses, _, thr = _fit_trace(vol, 5, 22, 20, "randomcube", 50, 0, 42, False)
    arrs["cube_vol"] = vol
    arrs["cube_losses"] = losses
    arrs["cube_thr"] = np.array([thr])
    for k, v in init.items():
        arrs["cube_init_" + k] = v
    for k, v in final.items():
        arrs["cube_final_" + k] = v
    # (ii) randompoint with the recorded index stream, non-cubic volume
    vol = make_volume((12, 20, 28), seed=43)
    nf, cf, side, init, final, losses, idx, thr = _fit_trace(vol, 4, 32, 20, "randompoint", 50, 1000, 42, True)
    arrs["pt_vol"] = vol
    arrs["pt_losses"] = losses
    arrs["pt_idx"] = idx.astype(np.int64)
    for k, v in init.items():
        arrs["pt_init_" + k] = v
    for k, v in final.items():
        arrs["pt_final_" + k] = v
    save("trace", **arrs)


# ----------------------------------------------------------------------------- 6. decode + metrics
def g_decode():
    arrs = {}
    vol = make_volume((16, 24, 40), seed=44)
    nf, cf, side, init, final, losses, _, thr = _fit_trace(vol, 5, 22, 20, "randomcube", 300, 0, 42, False)
    for k, v in final.items():
        arrs["net_" + k] = v
    arrs["vol"] = vol
    arrs["losses"] = losses
    arrs["side_min_max"] = np.array([side["min"], side["max"]], np.float64)
    arrs["side_nmin_nmax"] = np.array([side["normalized_min"], side["normalized_max"]], np.float64)
    dec = refmisc.reconstruct_flattened(list(vol.shape), 10000, nf.sample_nf, device="cpu", half=False,
                                        coords_mode=cf.Compress.coords_mode)
    arrs["dec_f32"] = dec.numpy().copy()
    out = refio.invnormalize_data(dec.clone(), side, **cf.Normalize)
    arrs["dec_u16"] = out
    v32, o32 = vol.astype(np.float32), out.astype(np.float32)
    arrs["mse"] = np.array([refmisc.cal_mse(v32, o32)], np.float64)
    arrs["psnr"] = np.array([refmisc.cal_psnr(v32, o32, 65535)], np.float64)
    arrs["ssim"] = np.array([refmisc.cal_ssim(v32, o32, 65535)], np.float64)
    # normalisation pins
    nd, ns = refio.normalize_data(vol, "minmaxany_0_100")
    arrs["norm_f32"] = nd.numpy()
    # truncation (not rounding) pin + clip pin
    probe = torch.tensor([[-3.0], [0.0], [49.99999], [50.0], [99.9999], [100.0], [140.0], [12.3456]])
    arrs["inv_probe_in"] = probe.numpy().copy()
    arrs["inv_probe_out"] = refio.invnormalize_data(probe.clone(), side, **cf.Normalize)
    # uint8 volume
    vol8 = (make_volume((8, 12, 16), seed=45).astype(np.float64) / 257.0).astype(np.uint8)
    nd8, ns8 = refio.normalize_data(vol8, "minmaxany_0_100")
    arrs["vol8"] = vol8
    arrs["norm8_f32"] = nd8.numpy()
    arrs["inv8"] = refio.invnormalize_data(nd8.clone() * 0.97 + 1.0, ns8, "minmaxany_0_100")
    # coords pins (torch.linspace is not lo + i*step)
    for n in (2, 3, 16, 63, 64, 100, 512):
        arrs["linspace_%d" % n] = torch.linspace(-1, 1, n).numpy()
    arrs["linspace01_37"] = torch.linspace(0, 1, 37).numpy()
    arrs["flat_coords_3_4_5"] = refdataset.create_flattened_coords((3, 4, 5), "-1,1").numpy()
    arrs["flat_coords_4_6"] = refdataset.create_flattened_coords((4, 6), "-1,1").numpy()
    # PSNR / SSIM on a second pair (noisy copy), incl. 2-D RGB uint8
    g = np.random.default_rng(3)
    a = make_volume((5, 40, 48), seed=46)
    b = np.clip(a.astype(np.int64) + g.integers(-300, 300, size=a.shape), 0, 65535).astype(np.uint16)
    arrs["pair_a"], arrs["pair_b"] = a, b
    arrs["pair_psnr"] = np.array([refmisc.cal_psnr(a.astype(np.float32), b.astype(np.float32), 65535)])
    arrs["pair_ssim"] = np.array([refmisc.cal_ssim(a.astype(np.float32), b.astype(np.float32), 65535)])
    arrs["pair_mse"] = np.array([refmisc.cal_mse(a.astype(np.float32), b.astype(np.float32))])
    img = g.integers(0, 256, size=(32, 36, 3)).astype(np.uint8)
    img2 = np.clip(img.astype(np.int64) + g.integers(-9, 9, size=img.shape), 0, 255).astype(np.uint8)
    arrs["img_a"], arrs["img_b"] = img, img2
    arrs["img_psnr"] = np.array([refmisc.cal_psnr(img.astype(np.float32), img2.astype(np.float32), 255)])
    arrs["img_ssim"] = np.array([refmisc.cal_ssim(img.astype(np.float32), img2.astype(np.float32), 255)])
    save("decode", **arrs)


# ----------------------------------------------------------------------------- 7. budget -> width
def g_budget():
    rows = []
    for L in range(3, 10):
        for cin, cout in ((3, 1), (2, 3), (2, 1)):
            for bytes_ in (100, 1000, 6687.5, 17924, 65536, 794628, 1e6, 7364612, 1e7):
                pc = bytes_ / 4.0
                F = SIREN.calc_features(pc, cin, cout, L, False)
                P = SIREN.calc_param_count(cin, cout, F, L, False)
                rows.append([L, cin, cout, bytes_, F, P])
    opt = load_opt()
    est = []
    for bytes_ in (6687.5, 17924.0, 794628.0, 2048.0):
        cf = copy.deepcopy(opt.CompressFramework)
        f, p, s = refmain.NFGR.estimate_module_size(bytes_, cf)
        est.append([bytes_, f, p, s])
    save("budget", table=np.array(rows, np.float64), estimate=np.array(est, np.float64))


# ----------------------------------------------------------------------------- 8. DivideTask host logic
def g_divide():
    arrs = {}
    vol = make_volume((12, 16, 20), seed=47)
    for dt in ("total_2_2_2", "every_5_8_7", "total_1_2_4"):
        chunks, _ = refmisc.divide_data(vol, dt)
        arrs["names_" + dt] = np.array([c["name"] for c in chunks])
        arrs["sizes_" + dt] = np.array([c["size"] for c in chunks])
        for alloc in ("equal", "by_size", "by_var", "by_d", "by_dv"):
            ch = refmisc.alloc_param(copy.deepcopy(chunks), 40000.0, alloc, 26)
            arrs["alloc_%s_%s_names" % (dt, alloc)] = np.array([c["name"] for c in ch])
            arrs["alloc_%s_%s" % (dt, alloc)] = np.array([c["param_size"] for c in ch], np.float64)
        # merge: paste chunks back (+1 so that a zero-filled gap would show)
        dl = [{"data": c["data"].copy(), "name": c["name"], "d": c["d"], "h": c["h"], "w": c["w"]} for c in chunks]
        arrs["merge_" + dt] = refmisc.merge_divided_data(dl, list(vol.shape))
    # drop rule: tiny budget makes small blocks fall below param_size_thres
    chunks, _ = refmisc.divide_data(vol, "every_11_16_20")
    ch = refmisc.alloc_param(copy.deepcopy(chunks), 300.0, "by_size", 26)
    arrs["drop_names"] = np.array([c["name"] for c in ch])
    arrs["drop_sizes"] = np.array([c["param_size"] for c in ch], np.float64)
    arrs["vol"] = vol
    arrs["feature_vol"] = np.array([cal_feature(vol)], np.float64)
    dn = []
    for (d, h, w, nb) in ((64, 512, 512, 4), (256, 256, 256, 8), (12, 16, 20, 6), (1024, 1024, 1024, 8), (64, 64, 64, 20), (30, 42, 70, 12)):
        dn.append([d, h, w, nb] + [int(v) for v in cal_divide_num(d, h, w, nb, 1e6)])
    arrs["divide_num"] = np.array(dn)
    arrs["divide_num_auto"] = np.array([int(v) for v in cal_divide_num(64, 64, 64, -1, 4 * 1361 * 9.5)])
    cps = {}
    for spec, ms in (("every_2000", 20000), ("every_3000", 10000), ("none", 500), ("100,400,900", 600)):  # int specs raise TypeError in the reference (misc.py:258)
        cps[str(spec) + "@" + str(ms)] = refmisc.parse_checkpoints(spec, ms)
    arrs["checkpoints_keys"] = np.array(list(cps.keys()))
    for i, k in enumerate(cps):
        arrs["checkpoints_%d" % i] = np.array(cps[k])
    v2 = vol.copy()
    v2[0, 0, 0, 0] = 65535
    for i, spec in enumerate((["value_65535_65535_1"], ["value_17000_20000_0.1"], ["quantile_16000_0.2_0.8_0.5"], ["exp_20000_0.5"], ["none"])):
        arrs["weight_%d" % i] = refmisc.parse_weight(v2, spec)
    arrs["weight_vol"] = v2
    save("divide", **arrs)




# ----------------------------------------------------------------------------- 9. deblock (reference deblock.py)
def g_deblock():
    """Runs the reference's Python deblocking filter (deblock.py: filter2d + its line enumeration) on a
    small blocky volume.  deblock.cpp (integer arithmetic, needs libtiff headers) cannot be built here."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_deblock", os.path.join(REF, "deblock.py"))
    rd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rd)
    rng = np.random.default_rng(21)
    d, h, w = 6, 24, 32
    base = make_volume((d, h, w), seed=48)[..., 0].astype(np.int64)
    names = []
    img = np.zeros((d, h, w), np.int64)
    for (z1, z2) in ((0, 2), (3, 5)):
        for (y1, y2) in ((0, 11), (12, 23)):
            for (x1, x2) in ((0, 15), (16, 31)):
                names.append("d_%d_%d-h_%d_%d-w_%d_%d" % (z1, z2, y1, y2, x1, x2))
                img[z1:z2 + 1, y1:y2 + 1, x1:x2 + 1] = base[z1:z2 + 1, y1:y2 + 1, x1:x2 + 1] + rng.integers(-150, 150)
    img = np.clip(img, 0, 65535).astype(np.uint16)[..., None]
    # drive the reference's own main() (deblock.py:79-130) on a scratch step directory: block names come from the directory
    # listing (sorted here: os.listdir order is file-system dependent and the order of the lines is part of the result), the
    # image I/O is replaced by in-memory arrays, filter2d is wrapped to record the boundary lines in processing order
    import tempfile
    step = tempfile.mkdtemp(prefix="brief_deblock_")
    os.makedirs(os.path.join(step, "decompressed"))
    open(os.path.join(step, "decompressed", "vol_decompressed.tif"), "wb").close()
    for n in names:
        os.makedirs(os.path.join(step, "compressed", "module", n))
    state = {}
    real_listdir, real_filter = os.listdir, rd.filter2d

    def run(index_a, index_b, thres):
        work = img.copy()
        lines, got = [], {}
        base = work.__array_interface__["data"][0]

        def rec_filter(pt, slice_img, *a):
            z = (slice_img.__array_interface__["data"][0] - base) // (work.shape[1] * work.shape[2] * work.shape[3] * work.itemsize)
            lines.append([int(z)] + [int(v) for v in pt])
            return real_filter(pt, slice_img, *a)
        rd.read_img = lambda path: work
        rd.save_img = lambda path, arr: got.setdefault("out", np.array(arr, copy=True))
        rd.filter2d = rec_filter
        os.listdir = lambda d: sorted(real_listdir(d))
        try:
            rd.main(step, index_a, index_b, thres)
        finally:
            os.listdir, rd.filter2d = real_listdir, real_filter
        return got["out"], lines
    out, lines = run(51, 2000, 65535)
    # a second run with a brightness threshold that disables part of the volume and a tighter beta
    out2, lines2 = run(48, 700, 20050)
    assert lines2 == lines
    names = sorted(names)
    import shutil
    shutil.rmtree(step, ignore_errors=True)
    save("deblock", img=img, names=np.array(names), lines=np.array(lines), out=out, out2=out2)


# ----------------------------------------------------------------------------- 10. Compress.half (fp16) mode
def g_half(modes=(("f32", False), ("f16", True)), out_name="half"):
    """The reference's own low-precision mode (main.py:388-399: module.half() for forward/backward, .float() for the
    optimizer step) against its fp32 mode from the same seed on the same volume: loss traces and end-of-fit PSNR.
    This anchors the band the MI355X low-precision path (bf16 MFMA, fp32 master weights) is held to."""
    arrs = {}
    vol = make_volume((24, 32, 40), seed=44)
    steps, L_, F_ = 3000, 5, 128            # a wide-ish net to the end of a 3000-step fit (SURVEY Appendix F's horizon)
    for tag, half in modes:
        opt = load_opt()
        cf = opt.CompressFramework
        cf.Compress.gpu = False
        cf.Decompress.gpu = False
        cf.Module.phi.layers = L_
        cf.Module.phi.w0 = 20
        cf.Compress.sampler.name = "randomcube"
        cf.Compress.half = False           # budget rule of the fp32 run for both (same net), the half flag only drives the loop
        refmain.reproduc(opt.Reproduc)
        nf = refmain.NFGR(cf)
        nf.device = "cpu"
        weight = refmisc.parse_weight(vol, cf.Compress.loss.weight)
        data, side = refio.normalize_data(vol, **cf.Normalize)
        feats, _ = nf.prepare_module(4.0 * SIREN.calc_param_count(3, 1, F_, L_, False))
        assert feats == F_
        if tag == "f32" and out_name == "half":
            for k, v in state_arrays(nf.module["phi"]).items():
                arrs["init_" + k] = v
        sampler = refmain.RandomCubeSampler(data, weight, cf.Compress.coords_mode, cf.Compress.sampler.cube_count,
                                            copy.deepcopy(cf.Compress.sampler.cube_len), steps, "cpu", True)
        optim = refmisc.configure_optimizer(nf.module["phi"].parameters(), cf.Compress.optimizer_name_phi, cf.Compress.lr_phi)
        sched = refmisc.configure_lr_scheduler(optim, cf.Compress.lr_scheduler_phi)
        thr, _ = refio.normalize_data(np.array(cf.Compress.loss.weight_thres), **cf.Normalize, max=side["max"], min=side["min"])
        thr = float(thr)
        losses = []
        phi = nf.module["phi"]
        for c, d, w in sampler:           # loop body exactly as main.py:385-400
            optim.zero_grad()
            if half:
                phi.half()
                d = d.half()
                yhat = phi.forward(c.half())
            else:
                yhat = phi.forward(c)
            loss = nf.loss_func(d, yhat, w, thr)
            loss.backward()
            if half:
                phi.float()
            optim.step()
            sched.step()
            losses.append(float(loss.item()))
        dec = refmisc.reconstruct_flattened(list(vol.shape), 10000, nf.sample_nf, device="cpu", half=False,
                                            coords_mode=cf.Compress.coords_mode)
        out = refio.invnormalize_data(dec.clone(), side, **cf.Normalize)
        arrs[tag + "_losses"] = np.array(losses, np.float64)
        arrs[tag + "_psnr"] = np.array([refmisc.cal_psnr(vol.astype(np.float32), out.astype(np.float32), 65535)], np.float64)
        print(tag, "loss[0,-1] =", losses[0], losses[-1], "psnr", arrs[tag + "_psnr"])
    if out_name == "half":
        arrs["vol"] = vol
    arrs["cfg"] = np.array([L_, F_, 20, steps])
    save(out_name, **arrs)


def g_half_self():
    """the reference against ITSELF on the case of half.npz: the same fp32 fit with another thread count (only the
    reduction order of its GEMMs changes).  Run as  BRIEF_GOLDEN_THREADS=1 python make_golden.py half_self  when
    half.npz was made with 4 threads: the gap between the two traces is the floor below which parity is meaningless."""
    g_half(modes=(("f32", False),), out_name="half_self")


def g_endfit_spread(runs=None):
    """How noisy is the REFERENCE's own end of fit on the case of half.npz?  The same fp32 fit, run k with ONE initial weight
    moved by one ulp (the perturbation rule of tools/endfit_spread.py: numpy default_rng(0) picks layer, element and direction);
    run 0 is the unperturbed fit (== half.npz / half_self.npz, depending on the thread count).  Stores the PSNR at the last step
    and the minimum / median / mean of the last 200 losses per run: the distribution the HIP path's runs are compared with
    (tests/test_gpu_parity.py::test_end_of_fit_distribution_matches_the_reference).
        BRIEF_GOLDEN_THREADS=4 python tests/golden/make_golden.py endfit_spread      (about 10 minutes per run on 4 threads)"""
    runs = int(os.environ.get("BRIEF_ENDFIT_RUNS", "6")) if runs is None else runs
    vol = make_volume((24, 32, 40), seed=44)
    steps, L_, F_ = 3000, 5, 128
    rng = np.random.default_rng(0)
    psnrs, mins, meds, means, picks = [], [], [], [], []
    for k in range(runs):
        opt = load_opt()
        cf = opt.CompressFramework
        cf.Compress.gpu = False
        cf.Decompress.gpu = False
        cf.Module.phi.layers = L_
        cf.Module.phi.w0 = 20
        cf.Compress.sampler.name = "randomcube"
        cf.Compress.half = False
        refmain.reproduc(opt.Reproduc)
        nf = refmain.NFGR(cf)
        nf.device = "cpu"
        weight = refmisc.parse_weight(vol, cf.Compress.loss.weight)
        data, side = refio.normalize_data(vol, **cf.Normalize)
        feats, _ = nf.prepare_module(4.0 * SIREN.calc_param_count(3, 1, F_, L_, False))
        assert feats == F_
        phi = nf.module["phi"]
        pick = (-1, -1, 0)
        if k > 0:
            l = int(rng.integers(0, L_))
            w = phi.net[l][0].weight.data.numpy().ravel()            # a view: the assignment below edits the parameter in place
            i = int(rng.integers(0, w.size))
            up = bool(rng.random() < 0.5)
            w[i] = np.nextafter(w[i], np.float32(1e9) if up else np.float32(-1e9))
            pick = (l, i, 1 if up else -1)
        sampler = refmain.RandomCubeSampler(data, weight, cf.Compress.coords_mode, cf.Compress.sampler.cube_count,
                                            copy.deepcopy(cf.Compress.sampler.cube_len), steps, "cpu", True)
        optim = refmisc.configure_optimizer(phi.parameters(), cf.Compress.optimizer_name_phi, cf.Compress.lr_phi)
        sched = refmisc.configure_lr_scheduler(optim, cf.Compress.lr_scheduler_phi)
        thr, _ = refio.normalize_data(np.array(cf.Compress.loss.weight_thres), **cf.Normalize, max=side["max"], min=side["min"])
        thr = float(thr)
        losses = []
        for c, d, w_ in sampler:           # loop body exactly as main.py:385-400
            optim.zero_grad()
            yhat = phi.forward(c)
            loss = nf.loss_func(d, yhat, w_, thr)
            loss.backward()
            optim.step()
            sched.step()
            losses.append(float(loss.item()))
        dec = refmisc.reconstruct_flattened(list(vol.shape), 10000, nf.sample_nf, device="cpu", half=False, coords_mode=cf.Compress.coords_mode)
        out = refio.invnormalize_data(dec.clone(), side, **cf.Normalize)
        ps = float(refmisc.cal_psnr(vol.astype(np.float32), out.astype(np.float32), 65535))
        tail = np.array(losses[-200:], np.float64)
        psnrs.append(ps); mins.append(tail.min()); meds.append(np.median(tail)); means.append(tail.mean()); picks.append(pick)
        print("reference run %d (perturbed layer, element, direction = %s): last-200 loss min %.5f median %.5f mean %.5f  PSNR %.3f dB" % (k, pick, tail.min(), np.median(tail), tail.mean(), ps), flush=True)
        save("endfit_spread", psnr=np.array(psnrs), last200_min=np.array(mins), last200_median=np.array(meds), last200_mean=np.array(means),
             picks=np.array(picks, np.int64), cfg=np.array([L_, F_, 20, steps]), threads=np.array([torch.get_num_threads()]))


# ----------------------------------------------------------------------------- 11. windowed RandomCubeSampler
def g_cube():
    """RandomCubeSampler with windows smaller than the volume (main.py:38-125): pop_size windows by unfold, cube_count
    windows per step drawn with torch.randint on the global CPU generator.  Records the window draws, the flat voxel
    indices every step touches (recovered from the sampled coordinates' positions), and the loss trace."""
    arrs = {}
    vol = make_volume((12, 20, 28), seed=49)
    steps, cube_len, cube_count = 30, [6, 8, 10], 3
    opt = load_opt()
    cf = opt.CompressFramework
    cf.Compress.gpu = False
    cf.Module.phi.layers = 4
    cf.Module.phi.w0 = 20
    cf.Compress.sampler.name = "randomcube"
    cf.Compress.sampler.cube_len = list(cube_len)
    cf.Compress.sampler.cube_count = cube_count
    refmain.reproduc(opt.Reproduc)
    nf = refmain.NFGR(cf)
    nf.device = "cpu"
    weight = refmisc.parse_weight(vol, cf.Compress.loss.weight)
    data, side = refio.normalize_data(vol, **cf.Normalize)
    feats, _ = nf.prepare_module(4.0 * SIREN.calc_param_count(3, 1, 32, 4, False))
    assert feats == 32
    for k, v in state_arrays(nf.module["phi"]).items():
        arrs["init_" + k] = v
    # voxel identity travels through the sampler as a "data" channel: flat index as float32 (exact below 2^24)
    ident = torch.arange(vol.size, dtype=torch.float32).reshape(vol.shape)
    s_id = refmain.RandomCubeSampler(ident, weight, cf.Compress.coords_mode, cube_count, list(cube_len), steps, "cpu", True)
    sampler = refmain.RandomCubeSampler(data, weight, cf.Compress.coords_mode, cube_count, list(cube_len), steps, "cpu", True)
    arrs["pop_size"] = np.array([sampler.pop_size])
    optim = refmisc.configure_optimizer(nf.module["phi"].parameters(), cf.Compress.optimizer_name_phi, cf.Compress.lr_phi)
    sched = refmisc.configure_lr_scheduler(optim, cf.Compress.lr_scheduler_phi)
    thr, _ = refio.normalize_data(np.array(cf.Compress.loss.weight_thres), **cf.Normalize, max=side["max"], min=side["min"])
    thr = float(thr)
    draws, vox, losses = [], [], []
    orig_randint = torch.randint

    def rec(*a, **k):
        r = orig_randint(*a, **k)
        draws.append(r.numpy().copy())
        return r
    torch.randint = rec
    try:
        for c, d, w in sampler:
            vox.append(s_id.data_cubes[torch.from_numpy(draws[-1]), :].reshape(-1).numpy().astype(np.int64))
            optim.zero_grad()
            yhat = nf.module["phi"].forward(c)
            loss = nf.loss_func(d, yhat, w, thr)
            loss.backward()
            optim.step()
            sched.step()
            losses.append(float(loss.item()))
    finally:
        torch.randint = orig_randint
    arrs["vol"] = vol
    arrs["cfg"] = np.array([4, 32, 20, steps, cube_count] + cube_len)
    arrs["draws"] = np.stack(draws).astype(np.int64)
    arrs["voxels"] = np.stack(vox)
    arrs["losses"] = np.array(losses, np.float64)
    for k, v in state_arrays(nf.module["phi"]).items():
        arrs["final_" + k] = v
    save("cube", **arrs)


# ----------------------------------------------------------------------------- 12. adaptive partition: the ILP's inputs
class _RecVar:
    def __init__(self, model, name):
        self.model, self.name, self.idx = model, name, len(model.vars)
        self.removed = False
        self.x = 0.0

    def _lin(self):
        return _RecLin({self.idx: 1.0})

    def __mul__(self, k):
        return _RecLin({self.idx: float(k)})
    __rmul__ = __mul__

    def __truediv__(self, k):
        return _RecLin({self.idx: 1.0 / float(k)})

    def __eq__(self, rhs):
        return ("==", self._lin(), float(rhs))

    def __le__(self, rhs):
        return ("<=", self._lin(), float(rhs))
    __hash__ = object.__hash__


class _RecLin:
    def __init__(self, coef):
        self.coef = dict(coef)

    def __mul__(self, k):
        return _RecLin({i: v * float(k) for i, v in self.coef.items()})
    __rmul__ = __mul__

    def __truediv__(self, k):
        return _RecLin({i: v / float(k) for i, v in self.coef.items()})

    def __eq__(self, rhs):
        return ("==", self, float(rhs))

    def __le__(self, rhs):
        return ("<=", self, float(rhs))
    __hash__ = object.__hash__


class _RecModel:
    """records what utils/adaptive_blocking.py hands to gurobipy (variables, objective, constraints) and solves the
    recorded binary program with scipy's HiGHS MILP solver, so that get_active()/draw() of the reference run."""

    def __init__(self):
        self.vars, self.cons, self.obj = [], [], None
        self.objVal = None

    def addVar(self, vtype=None, name=""):
        v = _RecVar(self, name)
        self.vars.append(v)
        return v

    def remove(self, v):
        v.removed = True

    def update(self):
        pass

    def setObjective(self, expr, sense):
        self.obj = expr

    def addConstr(self, c):
        self.cons.append(c)

    def optimize(self):
        from scipy.optimize import Bounds, LinearConstraint, milp
        n = len(self.vars)
        c = np.zeros(n)
        for i, v in self.obj.coef.items():
            c[i] = -v
        A, lo, hi = [], [], []
        for kind, lin, rhs in self.cons:
            row = np.zeros(n)
            for i, v in lin.coef.items():
                row[i] = v
            A.append(row)
            lo.append(rhs if kind == "==" else -np.inf)
            hi.append(rhs)
        ub = np.array([0.0 if v.removed else 1.0 for v in self.vars])
        res = milp(c, constraints=LinearConstraint(np.array(A), lo, hi), integrality=np.ones(n), bounds=Bounds(0, ub))
        assert res.status == 0, res.message
        for v, x in zip(self.vars, res.x):
            v.x = float(round(x))
        self.objVal = -float(res.fun)


def _rec_quicksum(items):
    tot = {}
    for it in items:
        lin = it._lin() if isinstance(it, _RecVar) else it
        for i, v in lin.coef.items():
            tot[i] = tot.get(i, 0.0) + v
    return _RecLin(tot)


def g_adaptive():
    """utils/adaptive_blocking.py:199-423 driven with a RECORDING stand-in for the gurobipy module (the solver is
    licensed software that is not in the image): the octree, its pruned-node set, every node's cal_feature and the
    binary program the reference builds (objective coefficients, constraint rows) are the reference's own; only the
    final argmax is computed by scipy's HiGHS instead of Gurobi.  Any optimal solution has the same objective value."""
    import utils.adaptive_blocking as rab
    gp = sys.modules["gurobipy"]
    gp.Model = _RecModel
    gp.quicksum = _rec_quicksum
    gp.GRB = types.SimpleNamespace(BINARY="B", MAXIMIZE=-1)
    arrs = {}
    cases = {
        # tag: (shape, seed, zero-box, var_thr, e_thr, Nb)
        "a": ((32, 32, 32), 50, None, 0, 0, 8),
        "b": ((32, 32, 32), 51, (slice(0, 16), slice(0, 16), slice(0, 32)), 0, 0, 12),      # a quarter of the volume is zero -> pruned
        "c": ((32, 64, 64), 52, (slice(16, 32), slice(32, 64), slice(32, 64)), 0, 0, 20),
        "d": ((64, 64, 64), 53, (slice(0, 32), slice(0, 64), slice(0, 64)), 0, 0, 64),
    }
    for tag, (shape, seed, zbox, vthr, ethr, Nb) in cases.items():
        vol = make_volume(shape, seed=seed)
        if zbox is not None:
            vol[zbox] = 0
        data = vol                       # (d,h,w,1): cal_feature takes the 4-D branch (3-D FFT), as adaptive_cal_tree passes it
        import math as _m
        minl = _m.floor(_m.log(Nb, 8))          # adaptive_cal_tree: utils/adaptive_blocking.py:400-401
        maxl = minl + 2
        tree = rab.OctTree(data, maxl, vthr, ethr)
        tree.solve_optim(Nb, minl)
        nodes = [[p.level, p.orderz, p.ordery, p.orderx, int(p.prune)] for p in tree.patch_list]
        feats = [0.0 if p.prune else float(p.feature) for p in tree.patch_list]
        act = [[p.level, p.orderz, p.ordery, p.orderx] for p in tree.get_active()]
        m = tree.optim_model
        coef = np.zeros(len(m.vars))
        for i, v in m.obj.coef.items():
            coef[i] = v
        arrs[tag + "_cfg"] = np.array(list(shape) + [seed, vthr, ethr, Nb, minl, maxl])
        if zbox is not None:
            arrs[tag + "_zbox"] = np.array([[s.start, s.stop] for s in zbox])
        arrs[tag + "_nodes"] = np.array(nodes, np.int64)           # pre-order (tree2list), the order of the variables
        arrs[tag + "_features"] = np.array(feats, np.float64)
        arrs[tag + "_objcoef"] = np.array([coef[p.active.idx] for p in tree.patch_list])      # in patch_list order
        arrs[tag + "_active"] = np.array(act, np.int64)
        arrs[tag + "_objval"] = np.array([m.objVal])
        arrs[tag + "_ncons"] = np.array([len(m.cons)])
        print(tag, "nodes", len(nodes), "pruned", int(sum(n[4] for n in nodes)), "active", len(act), "obj", m.objVal)
    save("adaptive", **arrs)


if __name__ == "__main__":
    which = sys.argv[1:] or ["init", "forward", "grads", "optim", "trace", "decode", "budget", "divide", "deblock", "half", "cube", "adaptive"]
    for w in which:
        globals()["g_" + w]()
