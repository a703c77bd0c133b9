"""GPU parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the same
seeded inputs and against the golden vectors produced by the reference.

Tolerances (fp32 path): forward <= 2e-5 of max|y|, every gradient tensor
<= 1e-4 of its max-abs, loss <= 1e-5 relative, 50-step loss trace <= 1e-4 relative,
optimizer update bit-exact, integer outputs bit-exact given identical yhat.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from brief_pytorch_amd import _lib
from brief_pytorch_amd.networks import SIREN
from oracle import oracle as O

from . import _bands

pytestmark = pytest.mark.gpu
DEV = "cuda"


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def make_net(L, F, w0, cin=3, cout=1, oa=False, seed=0, ws=None, bs=None):
    torch.manual_seed(seed)
    m = SIREN(coords_channel=cin, data_channel=cout, features=F, layers=L, w0=w0, output_act=oa)
    if ws is not None:
        for l in range(L):
            m.net[l][0].weight.data = torch.from_numpy(ws[l])
            m.net[l][0].bias.data = torch.from_numpy(bs[l])
    d = O.make_desc(cin, cout, L, F, w0, 30.0, oa)
    p = m.params.numpy().copy()
    return m.to(DEV), d, p


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e", "f"])
def test_forward_golden(golden, tag):
    g = golden("forward")
    L, F, w0, cin, cout, oa = [int(v) for v in g[tag + "_cfg"]]
    if tag + "_w0" in g:
        m, d, p = make_net(L, F, w0, cin, cout, bool(oa), ws=[g["%s_w%d" % (tag, l)] for l in range(L)],
                           bs=[g["%s_b%d" % (tag, l)] for l in range(L)])
    else:
        m, d, p = make_net(L, F, w0, cin, cout, bool(oa), seed=int(g[tag + "_seed"][0]))
    y = m.forward(torch.from_numpy(g[tag + "_x"]).to(DEV)).cpu().numpy()
    assert relerr(y, g[tag + "_y"]) < 2e-5                      # vs the reference itself
    assert relerr(y, O.forward(d, p, g[tag + "_x"])) < 2e-5     # vs the oracle


@pytest.mark.parametrize("L,F,cin,cout,n", [(2, 16, 3, 1, 100), (3, 33, 3, 1, 31), (4, 96, 2, 3, 1000), (5, 130, 3, 1, 257),
                                             (6, 200, 3, 1, 65), (3, 224, 3, 3, 1), (5, 256, 3, 1, 4097), (9, 64, 3, 1, 513),
                                             (4, 300, 3, 1, 200), (3, 512, 3, 1, 100), (9, 512, 3, 1, 257),
                                             (4, 340, 3, 1, 300), (3, 480, 3, 1, 150),       # 11 / 15 tiles: three left-over tiles shared along K
                                             (4, 290, 3, 1, 260), (3, 430, 3, 1, 140), (4, 330, 2, 1, 100), (5, 130, 3, 1, 700)])      # partial last k-tile (1, 2, 2, 1 real steps) under every sharing mode
def test_forward_shapes_vs_oracle(L, F, cin, cout, n):
    m, d, p = make_net(L, F, 20.0, cin, cout, seed=L * 100 + F)
    x = np.random.default_rng(F).uniform(-1, 1, size=(n, cin)).astype(np.float32)
    y = m.forward(torch.from_numpy(x).to(DEV)).cpu().numpy()
    assert y.shape == (n, cout)
    assert relerr(y, O.forward(d, p, x)) < 2e-5


def _check_grads(m, d, grads_ref, tol=1e-4, grads_ref64=None):
    """every gradient tensor within tol of its max-abs.  grads_ref64 (the oracle's float64 instantiation on the same
    inputs) widens the band to 3x the largest distance between the oracle's OWN f32 and f64 answers over the net's tensors:
    where the reference arithmetic is that far from the exact gradient (deep one- or two-wide nets multiply every rounding
    error by w0 per layer), a tighter agreement between two f32 evaluation orders is not a property of either.
    Every call is recorded (tests/_bands.py): the audit at the end of the session counts the widened ones and caps their band."""
    gw, gb = O.unpack_params(d, grads_ref)
    mw, mb = O.unpack_params(d, m.grads.cpu().numpy())
    worst = max(max(relerr(mw[l], gw[l]), relerr(mb[l], gb[l])) for l in range(d.layers))
    plain, own = tol, None
    if grads_ref64 is not None:
        w64, b64 = O.unpack_params(d, grads_ref64)
        own = max(max(relerr(gw[l], w64[l]), relerr(gb[l], b64[l])) for l in range(d.layers))
        if 3.0 * own > tol:
            print("_check_grads: band widened from %.1e to %.1e for layers=%d features=%d (oracle f32 vs f64: %.1e; HIP vs oracle f32: %.1e)" %
                  (tol, 3.0 * own, d.layers, d.features, own, worst))
        tol = max(tol, 3.0 * own)
    _bands.record("grad", "L=%d F=%d cin=%d cout=%d" % (d.layers, d.features, d.cin, d.cout), plain, tol, own, worst)
    for l in range(d.layers):
        assert relerr(mw[l], gw[l]) < tol, ("weight", l)
        assert relerr(mb[l], gb[l]) < tol, ("bias", l)


@pytest.mark.parametrize("tag", ["mse_unit", "mse_unit_thr", "mse_w_thr", "sl1_w", "mse_256"])
def test_loss_grads_golden(golden, tag):
    g = golden("grads")
    L, F, w0, kind, thr, beta = g[tag + "_cfg"]
    L, F, kind = int(L), int(F), int(kind)
    if tag + "_w0" in g:
        m, d, p = make_net(L, F, w0, ws=[g["%s_w%d" % (tag, l)] for l in range(L)], bs=[g["%s_b%d" % (tag, l)] for l in range(L)])
    else:
        m, d, p = make_net(L, F, w0, seed=3)
    x, y, w = (torch.from_numpy(g[tag + k]).to(DEV) for k in ("_x", "_y", "_w"))
    loss, yhat = m.train_step(x.shape[0], y, coords=x, weights=w, loss=["datal2", "datasmoothl1"][kind], thr=float(thr),
                              beta=float(beta), want_yhat=True)
    assert abs(loss.item() - g[tag + "_loss"][0]) / abs(g[tag + "_loss"][0]) < 1e-5
    assert relerr(yhat.cpu().numpy(), g[tag + "_yhat"]) < 2e-5
    lo, go, _, _ = O.loss_grad(d, p, g[tag + "_x"], g[tag + "_y"], g[tag + "_w"], kind, float(thr), float(beta))
    assert abs(loss.item() - lo) / abs(lo) < 1e-5
    _check_grads(m, d, go)
    if tag + "_gw0" in g and tag + "_w0" in g:      # reference's own gradients
        mw, mb = O.unpack_params(d, m.grads.cpu().numpy())
        for l in range(L):
            assert relerr(mw[l], g["%s_gw%d" % (tag, l)]) < 1e-4
            assert relerr(mb[l], g["%s_gb%d" % (tag, l)]) < 1e-4
    if tag == "mse_256":
        mw, mb = O.unpack_params(d, m.grads.cpu().numpy())
        assert relerr(mw[2][5], g["mse_256_gw2_row5"]) < 1e-4
        assert relerr(mb[1], g["mse_256_gb1"]) < 1e-4
        assert relerr(mw[0], g["mse_256_gw0"]) < 1e-4
        assert relerr(mw[4], g["mse_256_gw4"]) < 1e-4


@pytest.mark.parametrize("L,F,cin,cout,n,oa", [(2, 16, 3, 1, 100, False), (3, 33, 3, 1, 31, False), (4, 96, 2, 3, 1000, False),
                                                (5, 130, 3, 1, 2500, False), (6, 200, 3, 1, 650, False), (3, 224, 3, 3, 1, False),
                                                (5, 256, 3, 1, 9000, False), (9, 64, 3, 1, 513, False), (4, 48, 3, 1, 777, True),
                                                (4, 300, 3, 1, 700, False), (3, 512, 2, 3, 300, False), (5, 512, 3, 1, 1500, False),
                                                # 11 / 15 tiles (three left-over tiles shared along K, 2 x 6 / 2 x 8 k_wgrad quadrants), 14 tiles (7-tile quadrants)
                                                (4, 340, 3, 1, 1200, False), (3, 480, 3, 2, 900, False), (4, 440, 2, 1, 700, False),
                                                # a partial last k-tile (37 / 54 / 42 / 87 steps) under two, two, three and two shared left-over tiles
                                                (4, 290, 3, 1, 600, False), (3, 430, 3, 1, 500, False), (4, 330, 2, 1, 400, False), (3, 690, 3, 1, 300, False)])
def test_train_step_shapes_vs_oracle(L, F, cin, cout, n, oa):
    m, d, p = make_net(L, F, 20.0, cin, cout, oa, seed=L * 10 + F)
    rng = np.random.default_rng(F + n)
    x = rng.uniform(-1, 1, size=(n, cin)).astype(np.float32)
    y = rng.uniform(0, 100, size=(n, cout)).astype(np.float32)
    w = np.where(rng.uniform(size=(n, cout)) < 0.5, 0.25, 1.0).astype(np.float32)
    loss, _ = m.train_step(n, torch.from_numpy(y).to(DEV), coords=torch.from_numpy(x).to(DEV),
                           weights=torch.from_numpy(w).to(DEV), thr=30.0)
    lo, go, _, _ = O.loss_grad(d, p, x, y, w, 0, 30.0, 0.01)
    assert abs(loss.item() - lo) / abs(lo) < 1e-5
    _check_grads(m, d, go)


def test_random_configurations_vs_oracle():
    """a seeded random walk over the configuration space (tools/fuzz_parity.py runs the same generator for longer): layers 2..11,
    widths around every tile boundary, cin 2/3, cout 1..4, output activation, both losses, weight maps, thresholds, batch sizes
    1..6000 including every 32-multiple neighbourhood.  Forward, loss and every gradient tensor against the oracle."""
    rng = np.random.default_rng(2024)
    widths = list(range(1, 65)) + [65, 95, 96, 97, 127, 128, 129, 160, 191, 192, 200, 223, 224, 255, 256, 257, 300, 383, 384, 385, 450, 511, 512]
    for case in range(80):
        L = int(rng.integers(2, 12))
        F = int(rng.choice(widths))
        if F > 256 and L > 6:
            L = int(rng.integers(2, 7))
        cin, cout = int(rng.choice([2, 3])), int(rng.choice([1, 1, 1, 2, 3, 4]))
        oa = bool(rng.random() < 0.15)
        w0 = float(rng.choice([10.0, 20.0, 30.0]))
        n = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 255, 257, 1000, 2049, int(rng.integers(1, 6000))]))
        kind, thr, beta = int(rng.integers(0, 2)), float(rng.choice([0.0, 30.0, 200.0])), float(rng.choice([0.01, 1.0, 20.0]))
        use_w = bool(rng.random() < 0.6)
        m, d, p = make_net(L, F, w0, cin, cout, oa, seed=case)
        x = rng.uniform(-1, 1, size=(n, cin)).astype(np.float32)
        y = (rng.uniform(-1, 1, size=(n, cout)) if oa else rng.uniform(0, 100, size=(n, cout))).astype(np.float32)
        w = np.where(rng.uniform(size=(n, cout)) < 0.5, 0.25, 1.0).astype(np.float32) if use_w else np.ones((n, cout), np.float32)
        tag = (case, L, F, cin, cout, oa, w0, n, kind, thr, beta, use_w)
        assert relerr(m.forward(torch.from_numpy(x).to(DEV)).cpu().numpy(), O.forward(d, p, x)) < 2e-5, tag
        loss, _ = m.train_step(n, torch.from_numpy(y).to(DEV), coords=torch.from_numpy(x).to(DEV), weights=torch.from_numpy(w).to(DEV) if use_w else None,
                               loss=["datal2", "datasmoothl1"][kind], thr=thr, beta=beta)
        lo, go, _, _ = O.loss_grad(d, p, x, y, w, kind, thr, beta)
        # (a loss that is a small difference of O(1) numbers - one sample, sine head - is only as exact as yhat is)
        assert abs(loss.item() - lo) <= 1e-5 * max(abs(lo), 1e-2 * float(np.mean(y.astype(np.float64) ** 2))), tag
        gw, gb = O.unpack_params(d, go)
        mw, mb = O.unpack_params(d, m.grads.cpu().numpy())
        gmax = max(float(np.max(np.abs(t))) for t in list(gw) + list(gb))
        for l in range(L):
            for got, ref in ((mw[l], gw[l]), (mb[l], gb[l])):
                scale = max(float(np.max(np.abs(ref))), 1e-6 * gmax, 1e-30)
                assert float(np.max(np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)))) / scale < 1e-4, (tag, l)


@pytest.mark.parametrize("L,F,cin,cout,n", [(2, 8, 3, 1, 33), (3, 22, 3, 1, 4000), (4, 32, 2, 3, 129), (5, 22, 3, 1, 70000), (6, 17, 3, 1, 555),
                                             (7, 30, 3, 2, 3000), (8, 9, 3, 1, 64), (9, 32, 3, 1, 2049), (3, 64, 3, 1, 50000), (4, 35, 3, 1, 1000),
                                             (5, 56, 2, 1, 40000), (6, 40, 3, 4, 200), (7, 56, 3, 1, 66000), (8, 64, 3, 1, 321), (9, 33, 3, 1, 1500),
                                             (10, 24, 3, 1, 500), (11, 48, 3, 1, 500)])
def test_narrow_nets_on_chip_path_vs_oracle(L, F, cin, cout, n):
    """F <= 64 nets train through k_small (no stash; z in registers, dW accumulated in registers across the
    persistent tile loop): every (tile count, hidden-layer bucket) combination, ragged batches, enough samples
    that a workgroup walks several tiles; layers > 9 take the general path.  Same oracle, same tolerance."""
    m, d, p = make_net(L, F, 20.0, cin, cout, False, seed=L * 100 + F)
    rng = np.random.default_rng(F * 7 + n)
    x = rng.uniform(-1, 1, size=(n, cin)).astype(np.float32)
    y = rng.uniform(0, 100, size=(n, cout)).astype(np.float32)
    w = np.where(rng.uniform(size=(n, cout)) < 0.5, 0.25, 1.0).astype(np.float32)
    args = dict(coords=torch.from_numpy(x).to(DEV), weights=torch.from_numpy(w).to(DEV), thr=30.0)
    loss, _ = m.train_step(n, torch.from_numpy(y).to(DEV), **args)
    g1 = m.grads.clone()
    lo, go, _, _ = O.loss_grad(d, p, x, y, w, 0, 30.0, 0.01)
    assert abs(loss.item() - lo) / abs(lo) < 1e-5
    _check_grads(m, d, go)
    loss2, _ = m.train_step(n, torch.from_numpy(y).to(DEV), **args)
    assert torch.equal(g1, m.grads) and loss.item() == loss2.item()      # fixed summation order: bit-reproducible


def _random_cases(k, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(k):
        L = int(rng.integers(2, 11))
        F = int(rng.choice([rng.integers(1, 65), rng.integers(65, 257), rng.integers(257, 400)], p=[0.5, 0.35, 0.15]))
        cin = int(rng.choice([2, 3]))
        cout = int(rng.choice([1, 1, 2, 3, 4]))
        n = int(rng.choice([rng.integers(1, 40), rng.integers(40, 700), rng.integers(700, 4000)]))
        cases.append((L, F, cin, cout, n, bool(rng.integers(0, 4) == 0), str(rng.choice(["datal2", "datasmoothl1"])), bool(rng.integers(0, 2)), int(rng.integers(0, 3))))
    return cases


@pytest.mark.parametrize("L,F,cin,cout,n,oa,loss,use_w,thr_mode", _random_cases(40, 20261003))
def test_seeded_random_shapes_vs_oracle(L, F, cin, cout, n, oa, loss, use_w, thr_mode):
    """40 seeded random configurations (depth, width across all three train kernels' ranges, ragged batch sizes,
    2-D / 3-D coordinates, 1-4 channels, head sine, both losses, weights, weight threshold): forward, loss and every
    gradient against the oracle, and bit-reproducibility of the second run."""
    m, d, p = make_net(L, F, 20.0, cin, cout, oa, seed=L * 1000 + F + n)
    rng = np.random.default_rng(L * 7 + F * 3 + n)
    x = rng.uniform(-1, 1, size=(n, cin)).astype(np.float32)
    y = (rng.uniform(-1, 1, size=(n, cout)) if oa else rng.uniform(0, 100, size=(n, cout))).astype(np.float32)
    w = np.where(rng.uniform(size=(n, cout)) < 0.5, 0.25, 1.0).astype(np.float32) if use_w else None
    thr = [0.0, 30.0, -0.2][thr_mode] if not oa else [0.0, 0.3, -0.2][thr_mode]
    beta = 0.5 if loss == "datasmoothl1" else 0.01
    xt, yt = torch.from_numpy(x).to(DEV), torch.from_numpy(y).to(DEV)
    wt = torch.from_numpy(w).to(DEV) if use_w else None
    out = m.forward(xt).cpu().numpy()
    ref = O.forward(d, p, x)
    assert np.max(np.abs(out - ref)) <= 2e-5 * max(np.max(np.abs(ref)), 1e-3)
    l1, _ = m.train_step(n, yt, coords=xt, weights=wt, loss=loss, thr=thr, beta=beta)
    g1 = m.grads.clone()
    lo, go, _, _ = O.loss_grad(d, p, x, y, w if use_w else np.ones_like(y), 1 if loss == "datasmoothl1" else 0, thr, beta)
    assert abs(l1.item() - lo) <= 1e-5 * abs(lo) + 1e-9
    _, go64, _, _ = O.loss_grad(d, p, x, y, w if use_w else np.ones_like(y), 1 if loss == "datasmoothl1" else 0, thr, beta, f64=True)
    _check_grads(m, d, go, grads_ref64=go64)
    l2, _ = m.train_step(n, yt, coords=xt, weights=wt, loss=loss, thr=thr, beta=beta)
    assert torch.equal(g1, m.grads) and l1.item() == l2.item()


def test_train_step_is_deterministic_and_batch_split_linear():
    """Property tests at a larger size: two launches give identical bits (no atomics), and the
    gradient of a batch is the count-weighted sum of the gradients of its two halves."""
    m, d, p = make_net(5, 256, 20.0, seed=5)
    n = 20000
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)).to(DEV)
    y = torch.from_numpy(rng.uniform(0, 100, size=(n, 1)).astype(np.float32)).to(DEV)
    l1, _ = m.train_step(n, y, coords=x)
    g1 = m.grads.clone()
    l2, _ = m.train_step(n, y, coords=x)
    assert torch.equal(g1, m.grads) and l1.item() == l2.item()
    h = 12000
    m.train_step(h, y[:h].contiguous(), coords=x[:h].contiguous())
    ga = m.grads.clone().double()
    m.train_step(n - h, y[h:].contiguous(), coords=x[h:].contiguous())
    gb = m.grads.clone().double()
    comb = (ga * h + gb * (n - h)) / n
    assert relerr(comb.cpu().numpy(), g1.double().cpu().numpy()) < 2e-5


@pytest.mark.parametrize("name", ["Adamax", "Adam", "SGD"])
def test_optim_step_bit_exact_vs_oracle(name):
    rng = np.random.default_rng(11)
    n = 100003
    p = rng.normal(size=n).astype(np.float32)
    s1, s2 = np.zeros(n, np.float32), np.zeros(n, np.float32)
    tp, ts1, ts2 = (torch.from_numpy(a.copy()).to(DEV) for a in (p, s1, s2))
    for t in range(1, 6):
        g = (rng.normal(size=n) * 10 ** rng.uniform(-6, 2, size=n)).astype(np.float32)
        lr = 1e-3 * 0.2 ** (t // 3)
        O.optim_step(name, p, g, s1, s2, lr, t)
        tg = torch.from_numpy(g).to(DEV)
        _lib.check(_lib.lib().brief_optim_step(_lib.OPT_KIND[name], _lib.ptr(tp), _lib.ptr(tg), _lib.ptr(ts1), _lib.ptr(ts2), n,
                                               lr, 0.9, 0.999, 1e-8, t, _lib.stream_ptr()))
        assert np.array_equal(tp.cpu().numpy(), p), t
        assert np.array_equal(ts1.cpu().numpy(), s1) and np.array_equal(ts2.cpu().numpy(), s2)


def _fit_gpu(m, steps, vn, dims, idx_stream, thr, lr=1e-3):
    n_pop = int(np.prod(dims))
    tv = torch.from_numpy(vn.reshape(-1, 1)).to(DEV)
    s1, s2 = torch.zeros_like(m.params), torch.zeros_like(m.params)
    losses = []
    for t in range(1, steps + 1):
        if idx_stream is None:
            loss, _ = m.train_step(n_pop, tv, grid=(dims, -1.0, 1.0), thr=thr)
        else:
            idx = torch.from_numpy(idx_stream[t - 1]).to(DEV)
            loss, _ = m.train_step(idx.numel(), tv, idx=idx, grid=(dims, -1.0, 1.0), thr=thr)
        _lib.check(_lib.lib().brief_optim_step(0, _lib.ptr(m.params), _lib.ptr(m.grads), _lib.ptr(s1), _lib.ptr(s2), m.params.numel(),
                                               lr, 0.9, 0.999, 1e-8, t, _lib.stream_ptr()))
        m._stale = True
        losses.append(loss.item())
    return np.array(losses)


def test_trace_full_batch_golden(golden):
    g = golden("trace")
    m, d, p = make_net(5, 22, 20.0, ws=[g["cube_init_w%d" % l] for l in range(5)], bs=[g["cube_init_b%d" % l] for l in range(5)])
    vol = g["cube_vol"]
    vn, _ = O.normalize(vol)
    losses = _fit_gpu(m, 50, vn, vol.shape[:3], None, float(g["cube_thr"][0]))
    assert np.max(np.abs(losses - g["cube_losses"]) / g["cube_losses"]) < 1e-4
    for l in range(5):
        assert np.max(np.abs(m.net[l][0].weight.data.cpu().numpy() - g["cube_final_w%d" % l])) < 5e-5


def test_trace_randompoint_golden(golden):
    g = golden("trace")
    m, d, p = make_net(4, 32, 20.0, ws=[g["pt_init_w%d" % l] for l in range(4)], bs=[g["pt_init_b%d" % l] for l in range(4)])
    vol = g["pt_vol"]
    vn, side = O.normalize(vol)
    thr = float(O.normalize(np.array([65535], np.uint16), vmin=side["min"], vmax=side["max"])[0][0])
    losses = _fit_gpu(m, 50, vn, vol.shape[:3], g["pt_idx"], thr)
    assert np.max(np.abs(losses - g["pt_losses"]) / g["pt_losses"]) < 1e-4
    for l in range(4):
        assert np.max(np.abs(m.net[l][0].weight.data.cpu().numpy() - g["pt_final_w%d" % l])) < 5e-5


def test_trace_windowed_cube_golden(golden):
    """the reference's RandomCubeSampler with 3 windows of 6x8x10 per step on a 12x20x28 volume (main.py:38-125),
    recorded voxel stream replayed through the fused step: 30-step loss trace and final weights"""
    g = golden("cube")
    L = int(g["cfg"][0])
    m, d, p = make_net(L, int(g["cfg"][1]), float(g["cfg"][2]), ws=[g["init_w%d" % l] for l in range(L)], bs=[g["init_b%d" % l] for l in range(L)])
    vol = g["vol"]
    vn, side = O.normalize(vol)
    thr = float(O.normalize(np.array([65535], np.uint16), vmin=side["min"], vmax=side["max"])[0][0])
    losses = _fit_gpu(m, int(g["cfg"][3]), vn, vol.shape[:3], g["voxels"], thr)
    assert np.max(np.abs(losses - g["losses"]) / g["losses"]) < 1e-4
    for l in range(L):
        assert np.max(np.abs(m.net[l][0].weight.data.cpu().numpy() - g["final_w%d" % l])) < 5e-5


def _fit_half_golden(golden, precision, pick=None):
    """the volume, net and schedule of tests/golden/half.npz (5x128 SIREN, 24x32x40 volume, full batch, Adamax) fitted
    by brief_siren_fit; returns (loss trace, PSNR of the decoded uint16 volume).  pick = (layer, element, direction): that
    initial weight is moved by one ulp first (tests/golden/endfit_spread.npz records the reference's runs under the same picks)"""
    from brief_pytorch_amd.fit import Fitter
    g = golden("half")
    L, F, w0, steps = (int(v) for v in g["cfg"])
    vol = g["vol"]
    vn, side = O.normalize(vol)
    m = SIREN(features=F, layers=L, w0=w0, precision=precision)
    for l in range(L):
        w = g["init_w%d" % l].copy()
        if pick is not None and int(pick[0]) == l:
            flat = w.ravel()
            flat[int(pick[1])] = np.nextafter(flat[int(pick[1])], np.float32(1e9) if int(pick[2]) > 0 else np.float32(-1e9))
        m.net[l][0].weight.data = torch.from_numpy(w)
        m.net[l][0].bias.data = torch.from_numpy(g["init_b%d" % l])
    m.to(DEV)
    thr = float(O.normalize(np.array([65535], np.uint16), vmin=side["min"], vmax=side["max"])[0][0])
    tv = torch.from_numpy(vn.reshape(-1, 1)).to(DEV)
    fit = Fitter(m, tv, vol.shape[:3], sampler="full", optimizer="Adamax", lr=1e-3, thr=thr,
                 scheduler={"name": "MultiStepLR", "milestones": [50000, 60000, 70000], "gamma": 0.2})
    losses = fit.run(steps, log=True).cpu().numpy().astype(np.float64)
    dec = m.decode_grid(vol.shape[:3], out_kind="u16", scale=(0.0, 100.0), vrange=(side["min"], side["max"])).cpu().numpy().reshape(vol.shape)
    return g, losses, O.psnr(vol, dec, 65535)


def test_end_of_fit_3000_steps_golden(golden):
    """a 3000-step fit of a 128-wide net to its end against the reference's own runs.  The band is the reference's OWN
    noise on this case, measured: tests/golden/half.npz (4 CPU threads) and half_self.npz (1 thread: only the reduction
    order of its GEMMs changes) are 2e-7 apart at step 50, 9e-5 at step 100, 6e-4 at step 200, 1.4e-2 at step 500; the CPU
    oracle sits 8e-4 from the 4-thread trace at step 200.  End of fit: the loss of this case oscillates between 1.5 and
    3.0 over its last 200 steps, so the PSNR AT step 3000 is one sample of that oscillation: changing ONE initial weight
    by one ulp moves it over 55.18 ... 56.28 dB on one build and 54.15 ... 56.27 dB on another (tools/endfit_spread.py,
    profiles/r03_endfit_spread.md: std 0.3-0.7 dB; the reference's two runs: 56.26 / 56.42).  What IS stable is the
    envelope: the minimum of the last 200 losses stays within 1.536 ... 1.617 over all those runs (reference 1.550 /
    1.504), so the quality of the fit is pinned on the envelope and the PSNR at the last step gets the measured noise."""
    g, losses, psnr = _fit_half_golden(golden, "fp32")
    g2 = golden("half_self")
    ref, ref2 = g["f32_losses"], g2["f32_losses"]
    err = np.abs(losses - ref) / ref
    self_err = np.abs(ref2 - ref) / ref
    p1, p2 = float(g["f32_psnr"][0]), float(g2["f32_psnr"][0])
    env, env_ref = losses[-200:].min(), 0.5 * (ref[-200:].min() + ref2[-200:].min())
    med, med_ref = np.median(losses[-200:]), 0.5 * (np.median(ref[-200:]) + np.median(ref2[-200:]))
    print("3000-step fit: trace error through step 50 / 100 / 200: %.1e / %.1e / %.1e (reference vs itself: %.1e / %.1e / %.1e); "
          "last-200 loss min %.4f median %.4f (reference %.4f / %.4f); PSNR %.3f dB (reference %.3f and %.3f dB)"
          % (err[:50].max(), err[:100].max(), err[:200].max(), self_err[:50].max(), self_err[:100].max(), self_err[:200].max(),
             env, med, env_ref, med_ref, psnr, p1, p2))
    assert err[:50].max() < 1e-5
    assert err[:100].max() < 3e-4 and err[:200].max() < 2e-3          # ~3x the reference's own divergence at those steps
    assert abs(env - env_ref) < 0.08 * env_ref and abs(med - med_ref) < 0.12 * med_ref     # the envelope of the end of the fit
    assert abs(psnr - 0.5 * (p1 + p2)) < 2.5                           # one sample of the oscillation (see above)


def test_end_of_fit_distribution_matches_the_reference(golden):
    """Is the end of a fit on the HIP path as good as the reference's — not one sample of the last-step oscillation, but the distribution?
    tests/golden/endfit_spread.npz holds the REFERENCE's own 3000-step fits of the half.npz case, run k with one initial weight moved by one
    ulp (make_golden.py endfit_spread): its PSNR at the last step spreads over ~1 dB too (56.23 / 55.35 / 55.84 / ... dB), so the two runs of
    half.npz / half_self.npz (56.26, 56.42) were the top of that distribution, not its centre.  The same perturbed fits on the HIP path must
    give the same MEANS: PSNR within 0.5 dB, and the median / minimum of the last 200 losses (what the oscillation does not touch) within
    4 % / 3 % (round-3 advisor: a 2.5 dB band around single samples could hide a systematic 0.5 - 1 dB loss of the fract-based phase arithmetic)."""
    ref = golden("endfit_spread")
    runs = len(ref["psnr"])
    assert runs >= 4
    ps, mins, meds = [], [], []
    for k in range(runs):
        pick = None if int(ref["picks"][k][0]) < 0 else ref["picks"][k]
        _, losses, psnr = _fit_half_golden(golden, "fp32", pick=pick)
        ps.append(psnr); mins.append(losses[-200:].min()); meds.append(np.median(losses[-200:]))
    ps, mins, meds = np.array(ps), np.array(mins), np.array(meds)
    print("end of fit over %d one-ulp-perturbed runs: PSNR mean %.3f (std %.2f) against the reference's %.3f (std %.2f); last-200 loss median %.4f against %.4f, "
          "minimum %.4f against %.4f" % (runs, ps.mean(), ps.std(), ref["psnr"].mean(), ref["psnr"].std(), meds.mean(), ref["last200_median"].mean(),
                                          mins.mean(), ref["last200_min"].mean()))
    assert abs(ps.mean() - ref["psnr"].mean()) < 0.5
    assert abs(meds.mean() - ref["last200_median"].mean()) < 0.04 * ref["last200_median"].mean()
    assert abs(mins.mean() - ref["last200_min"].mean()) < 0.03 * ref["last200_min"].mean()
    assert ps.min() > ref["psnr"].min() - 1.5 and ps.max() < ref["psnr"].max() + 1.0


def test_decode_golden(golden):
    g = golden("decode")
    vol = g["vol"]
    m, d, p = make_net(5, 22, 20.0, ws=[g["net_w%d" % l] for l in range(5)], bs=[g["net_b%d" % l] for l in range(5)])
    dims = vol.shape[:3]
    dec = m.decode_grid(dims).cpu().numpy().reshape(vol.shape)
    assert relerr(dec, g["dec_f32"]) < 2e-5
    _, side = O.normalize(vol)
    u16 = m.decode_grid(dims, out_kind="u16", scale=(0.0, 100.0), vrange=(side["min"], side["max"])).cpu().numpy().reshape(vol.shape)
    # fused epilogue == oracle epilogue applied to the kernel's own yhat, bit for bit
    assert np.array_equal(u16, O.invnormalize(dec, side))
    diff = np.abs(u16.astype(np.int64) - g["dec_u16"].astype(np.int64))
    assert diff.max() <= 1 and (diff != 0).mean() < 0.02
    assert abs(O.psnr(vol, u16, 65535) - g["psnr"][0]) < 0.01
    # chunked / offset decode gives the same voxels
    part = m.decode_grid(dims, offset=1000, count=777).cpu().numpy()
    assert np.array_equal(part.ravel(), dec.ravel()[1000:1777])
    # in-kernel coordinate synthesis == explicit coords from the oracle's linspace restatement
    xy = torch.from_numpy(O.grid_coords(dims)).to(DEV)
    assert np.array_equal(m.forward(xy).cpu().numpy().ravel(), dec.ravel())


def test_decode_2d_u8():
    m, d, p = make_net(4, 40, 20.0, cin=2, cout=3, seed=9)
    dims = (37, 53)
    dec = m.decode_grid(dims).cpu().numpy()
    ref = O.forward(d, p, O.grid_coords(dims))
    assert relerr(dec, ref) < 2e-5
    side = {"dtype": "uint8", "min": 3.0, "max": 250.0}
    u8 = m.decode_grid(dims, out_kind="u8", scale=(0.0, 100.0), vrange=(3.0, 250.0)).cpu().numpy()
    assert np.array_equal(u8, O.invnormalize(dec * 400.0 / 400.0, side))


def test_sample_indices_and_sse():
    n, pop = 200000, 512 ** 3
    a = torch.empty(n, dtype=torch.int64, device=DEV)
    b = torch.empty(n, dtype=torch.int64, device=DEV)
    L = _lib.lib()
    _lib.check(L.brief_sample_indices(_lib.ptr(a), n, pop, 42, 7, _lib.stream_ptr()))
    _lib.check(L.brief_sample_indices(_lib.ptr(b), n, pop, 42, 7, _lib.stream_ptr()))
    assert torch.equal(a, b)
    _lib.check(L.brief_sample_indices(_lib.ptr(b), n, pop, 42, 8, _lib.stream_ptr()))
    assert not torch.equal(a, b)
    an = a.cpu().numpy()
    assert an.min() >= 0 and an.max() < pop
    assert abs(an.mean() / pop - 0.5) < 0.01 and len(np.unique(an)) > 0.99 * n
    hist = np.histogram(an, bins=16, range=(0, pop))[0]
    assert hist.min() > 0.9 * n / 16
    rng = np.random.default_rng(0)
    u = rng.integers(0, 65536, size=1 << 20).astype(np.uint16)
    v = rng.integers(0, 65536, size=1 << 20).astype(np.uint16)
    out = torch.zeros(1, dtype=torch.float64, device=DEV)
    tu, tv = torch.from_numpy(u).to(DEV), torch.from_numpy(v).to(DEV)
    _lib.check(L.brief_sse_u16(_lib.ptr(tu), _lib.ptr(tv), u.size, _lib.ptr(out), _lib.stream_ptr()))
    assert out.item() == float(((u.astype(np.int64) - v.astype(np.int64)) ** 2).sum())


def test_errors_are_loud():
    L = _lib.lib()
    d = _lib.SirenDesc(3, 1, 5, 5000, 20.0, 30.0, 0, 0)            # (widths up to 4096 features run: tests/test_gpu_wider.py)
    assert L.brief_packed_count(C.byref(d)) == -1 and b"4096" in L.brief_last_error()
    m, _, _ = make_net(3, 16, 20.0)
    with pytest.raises(_lib.BriefError):
        m.train_step(10, torch.zeros(10, 1, device=DEV), coords=None, grid=((4, 4), -1.0, 1.0))   # ndim != cin
    # a shape the library refuses surfaces as ITS message from the module, not as a torch allocation error on a count of -1
    from brief_pytorch_amd.networks import SIREN
    wide16 = SIREN(features=1100, layers=4, w0=20, precision="bf16").to(DEV)       # the bf16 path stops at 512 features
    with pytest.raises(_lib.BriefError, match="512"):
        wide16.sync_packed()


def test_deblock_vs_oracle_and_golden(golden):
    from brief_pytorch_amd.deblock import block_edges, deblock_volume
    g = golden("deblock")
    names = [str(n) for n in g["names"]]
    assert sum((e[1] - e[0] + 1) for e in block_edges(names)) == len(g["lines"])
    t = torch.from_numpy(g["img"].copy()).to(DEV)
    assert np.array_equal(deblock_volume(t, names).cpu().numpy(), g["out"])                 # reference deblock.py, bit-exact
    t2 = torch.from_numpy(g["img"].copy()).to(DEV)
    assert np.array_equal(deblock_volume(t2, names, 48, 700, 20050).cpu().numpy(), g["out2"])
    # larger volume, both arithmetic flavours against the oracle
    from brief_pytorch_amd.misc import divide_data
    from brief_pytorch_amd.synthetic import make_volume
    vol = make_volume((20, 96, 128), seed=8)
    rng = np.random.default_rng(2)
    chunks, _ = divide_data(vol, "total_2_3_4")
    blocky = vol.astype(np.int64)
    for c in chunks:
        r = c["d"], c["h"], c["w"]
        blocky[r[0][0]:r[0][1] + 1, r[1][0]:r[1][1] + 1, r[2][0]:r[2][1] + 1] += rng.integers(-200, 200)
    blocky = np.clip(blocky, 0, 65535).astype(np.uint16)
    bn = [c["name"] for c in chunks]
    for mode in (1, 0):
        tt = torch.from_numpy(blocky.copy()).to(DEV)
        assert np.array_equal(deblock_volume(tt, bn, mode=mode).cpu().numpy(), O.deblock(blocky, bn, mode=mode)), mode


@pytest.mark.parametrize("L,F,name", [(5, 256, "Adamax"), (4, 40, "Adam"), (3, 96, "SGD"), (2, 20, "Adamax"), (4, 400, "Adamax")])
def test_fit_step_equals_separate_calls(L, F, name):
    """brief_siren_fit_step (reduce + optimizer + packed write-through fused) is bit-identical to
    train_step + optim_step + repack, parameters AND the fragment-ordered copy."""
    n = 5000
    rng = np.random.default_rng(L + F)
    dims = (20, 25, 30)
    tv = torch.from_numpy(rng.uniform(0, 100, size=(int(np.prod(dims)), 1)).astype(np.float32)).to(DEV)
    idxs = [torch.from_numpy(rng.integers(0, int(np.prod(dims)), size=n)).to(DEV) for _ in range(3)]
    ma, _, _ = make_net(L, F, 20.0, seed=77)
    mb, _, _ = make_net(L, F, 20.0, seed=77)
    kind = _lib.OPT_KIND[name]
    sa1, sa2 = torch.zeros_like(ma.params), torch.zeros_like(ma.params)
    sb1, sb2 = torch.zeros_like(mb.params), torch.zeros_like(mb.params)
    for t in range(1, 4):
        la, _ = ma.train_step(n, tv, idx=idxs[t - 1], grid=(dims, -1.0, 1.0))
        la = la.clone()
        _lib.check(_lib.lib().brief_optim_step(kind, _lib.ptr(ma.params), _lib.ptr(ma.grads), _lib.ptr(sa1), _lib.ptr(sa2),
                                               ma.params.numel(), 1e-3, 0.9, 0.999, 1e-8, t, _lib.stream_ptr()))
        ma._stale = True
        ma.sync_packed()
        lb = mb.fit_step(n, tv, kind, sb1, sb2, 1e-3, t, idx=idxs[t - 1], grid=(dims, -1.0, 1.0))
        assert torch.equal(la, lb) and torch.equal(ma.grads, mb.grads)
        assert torch.equal(ma.params, mb.params) and torch.equal(sa1, sb1) and torch.equal(sa2, sb2)
        assert torch.equal(ma.packed, mb.packed)


def _mk_fitter(L, F, dims, sampler, n, seed, sched=None, opt="Adamax"):
    from brief_pytorch_amd.fit import Fitter
    torch.manual_seed(seed)
    m = SIREN(coords_channel=3, data_channel=1, features=F, layers=L, w0=20.0).to(DEV)
    pop = int(np.prod(dims))
    tv = (torch.rand(pop, 1, generator=torch.Generator().manual_seed(seed + 1)) * 100).to(DEV)
    return Fitter(m, tv, dims, sampler=sampler, sample_size=n, seed=seed, scheduler=sched, optimizer=opt)


@pytest.mark.parametrize("L,F,sampler,opt", [(5, 22, "full", "Adamax"), (4, 96, "randompoint", "Adam"), (3, 64, "randompoint", "SGD")])
def test_fit_many_steps_per_call_equals_single_steps(L, F, sampler, opt):
    """brief_siren_fit(steps) == steps x brief_siren_fit_step, bit for bit, including the MultiStepLR
    bookkeeping (a doubled milestone multiplies twice) and a call that starts in the middle of the schedule."""
    sched = {"name": "MultiStepLR", "milestones": [7, 15, 15, 30], "gamma": 0.5}
    dims = (16, 24, 20)
    a = _mk_fitter(L, F, dims, sampler, 3000, 11, sched, opt)
    b = _mk_fitter(L, F, dims, sampler, 3000, 11, sched, opt)
    assert torch.equal(a.m.params, b.m.params)
    ref = [float(a.step()) for _ in range(40)]
    log1 = b.run(12, log=True)
    log2 = b.run(28, log=True)
    got = torch.cat([log1, log2]).cpu().numpy()
    assert b.t == 40 and torch.equal(a.m.params, b.m.params) and torch.equal(a.s1, b.s1) and torch.equal(a.s2, b.s2)
    assert np.array_equal(np.asarray(ref, np.float32), got)
    assert float(b.m._loss) == ref[-1]
    assert torch.equal(a.m.packed, b.m.packed)


def test_steplr_schedule_matches_torch():
    """utils/misc.py:190-191 StepLR: lr_t = lr * gamma^((t-1) // step_size), also through run()"""
    sched = {"name": "StepLR", "step_size": 4, "gamma": 0.5}
    a = _mk_fitter(3, 32, (8, 8, 8), "full", 0, 3, sched)
    b = _mk_fitter(3, 32, (8, 8, 8), "full", 0, 3, sched)
    ref = torch.optim.lr_scheduler.StepLR(torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=1e-3), 4, 0.5)
    for t in range(1, 11):
        assert abs(a.lr_at(t) - ref.get_last_lr()[0]) < 1e-18
        ref.optimizer.step(); ref.step()
        a.step()
    log = b.run(10, log=True)
    assert b.t == 10 and log.shape == (10,) and torch.equal(a.m.params, b.m.params)


def test_cyclic_lr_and_index_stream_run_inside_the_cabi_loop():
    """brief_fit_job's lr_table / beta1_table (CyclicLR cycles Adam's beta1) and its device-resident index stream (windowed
    RandomCubeSampler, main.py:38-125): run(k) in one brief_siren_fit call == k x step(), bit for bit; a run that is cut into
    several calls by the index-stream memory budget too."""
    from brief_pytorch_amd.fit import Fitter
    from brief_pytorch_amd.framework import _CubeIndexStream
    sched = {"name": "CyclicLR", "base_lr": 1e-4, "max_lr": 2e-3, "step_size_up": 5, "step_size_down": 3, "mode": "triangular2"}
    a = _mk_fitter(4, 48, (8, 16, 16), "randompoint", 700, 21, sched, "Adam")
    b = _mk_fitter(4, 48, (8, 16, 16), "randompoint", 700, 21, sched, "Adam")
    ref = [float(a.step()) for _ in range(19)]
    got = torch.cat([b.run(8, log=True), b.run(11, log=True)]).cpu().numpy()
    assert np.array_equal(np.asarray(ref, np.float32), got)
    assert torch.equal(a.m.params, b.m.params) and torch.equal(a.s1, b.s1) and torch.equal(a.s2, b.s2) and torch.equal(a.m.packed, b.m.packed)
    # windowed cube sampler: the window draws come from a generator seeded alike on both sides
    dims, cl = (12, 20, 16), (5, 8, 6)
    fits = []
    for k in range(3):
        torch.manual_seed(5)
        m = SIREN(features=40, layers=4, w0=20.0).to(DEV)
        tv = (torch.rand(int(np.prod(dims)), 1, generator=torch.Generator().manual_seed(6)) * 100).to(DEV)
        st = _CubeIndexStream(dims, cl, 3, DEV, generator=torch.Generator().manual_seed(99))
        fits.append(Fitter(m, tv, dims, sampler="randompoint", sample_size=st.n, index_stream=st, scheduler={"name": "StepLR", "step_size": 3, "gamma": 0.7}))
    ref = [float(fits[0].step()) for _ in range(10)]
    got = fits[1].run(10, log=True).cpu().numpy()
    assert np.array_equal(np.asarray(ref, np.float32), got) and torch.equal(fits[0].m.params, fits[1].m.params)
    fits[2].INDEX_STREAM_BYTES = 8 * fits[2].n * 4              # four steps per call: 10 steps = 3 calls
    got3 = fits[2].run(10, log=True).cpu().numpy()
    assert fits[2].max_steps_per_call() == 4 and np.array_equal(got, got3) and torch.equal(fits[1].m.params, fits[2].m.params)
    # a job whose index pointer has no stride is refused
    j, _ = fits[1].job(2)
    j.idx_stride = 0
    import ctypes as C
    assert _lib.lib().brief_siren_fit(C.byref(j), 2, _lib.stream_ptr()) != 0 and b"idx_stride" in _lib.lib().brief_last_error()


def test_multi_fit_equals_individual_fits():
    """brief_multi_fit: blocks co-trained on internal streams give exactly the results of fitting each alone
    (10 jobs > 8 pool streams: two jobs share a stream)."""
    from brief_pytorch_amd.fit import MultiFitter
    shapes = [(5, 22, (16, 16, 16), "full", 0), (7, 56, (8, 32, 32), "randompoint", 2500), (3, 64, (16, 16, 16), "full", 0),
              (5, 256, (8, 32, 32), "randompoint", 3000), (4, 35, (8, 16, 16), "full", 0), (9, 30, (8, 16, 16), "randompoint", 1000),
              (2, 16, (8, 8, 8), "full", 0), (5, 130, (8, 16, 32), "randompoint", 2000), (6, 40, (8, 16, 16), "full", 0),
              (3, 300, (8, 16, 16), "randompoint", 1500)]
    solo = [_mk_fitter(L, F, d, s, n, 100 + i) for i, (L, F, d, s, n) in enumerate(shapes)]
    group = [_mk_fitter(L, F, d, s, n, 100 + i) for i, (L, F, d, s, n) in enumerate(shapes)]
    for f in solo:
        f.run(30)
    mf = MultiFitter(group)
    logs = mf.run(18, log=True)
    mf.run(12)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(solo, group)):
        assert b.t == 30 and torch.equal(a.m.params, b.m.params), shapes[i]
        assert float(a.m._loss) == float(b.m._loss)
        assert logs[i].shape == (18,) and torch.isfinite(logs[i]).all()


def test_cotrained_mid_and_wide_nets_equal_their_solo_fits_under_concurrency():
    """k_lean shares the tiles that nt % 4 leaves over along K and folds the partial tiles through LDS: a missing barrier there is a
    RACE that only shows when other kernels perturb the timing (tools/fuzz_multifit.py found one: a co-trained 130-wide net differed from
    its solo fit in one run of twenty).  Tile counts with one and two left-over tiles (5, 6, 9, 10, 13, 17, 18), co-trained on the
    library's stream pool next to narrow nets, three rounds: every fit bit-identical to its solo run."""
    from brief_pytorch_amd.fit import MultiFitter
    shapes = [(3, 130, (16, 8, 8), "full", 0), (4, 160, (8, 16, 16), "randompoint", 1500), (3, 192, (8, 16, 16), "full", 0),
              (3, 288, (8, 16, 16), "randompoint", 1000), (3, 320, (8, 8, 16), "full", 0), (3, 416, (8, 8, 16), "full", 0),
              (3, 527, (8, 8, 8), "full", 0), (3, 576, (8, 8, 8), "randompoint", 700), (5, 22, (16, 16, 16), "full", 0),
              (5, 22, (16, 16, 8), "full", 0), (3, 64, (16, 16, 16), "full", 0)]
    for rnd in range(3):
        solo = [_mk_fitter(L, F, d, s, n, 300 + 20 * rnd + i) for i, (L, F, d, s, n) in enumerate(shapes)]
        group = [_mk_fitter(L, F, d, s, n, 300 + 20 * rnd + i) for i, (L, F, d, s, n) in enumerate(shapes)]
        for f in solo:
            f.run(16)
        MultiFitter(group).run(16)
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(solo, group)):
            assert torch.equal(a.m.params, b.m.params), (rnd, shapes[i])


def test_fit_job_replays_an_index_stream_and_rejects_a_malformed_one():
    """a replayed per-step index stream (any callable t -> indices) runs inside brief_siren_fit as a device-resident
    [steps, n] tensor: same bits as step(); a stream of the wrong length or dtype is refused before the launch"""
    from brief_pytorch_amd._lib import BriefError
    gens = [torch.Generator().manual_seed(3) for _ in range(2)]
    fa, fb = (_mk_fitter(3, 16, (8, 8, 8), "randompoint", 100, 1) for _ in range(2))
    fa.index_stream = lambda t: torch.randint(0, 512, (100,), generator=gens[0]).to(DEV)
    fb.index_stream = lambda t: torch.randint(0, 512, (100,), generator=gens[1]).to(DEV)
    ref = [float(fa.step()) for _ in range(6)]
    got = fb.run(6, log=True).cpu().numpy()
    assert np.array_equal(np.asarray(ref, np.float32), got) and torch.equal(fa.m.params, fb.m.params)
    fb.index_stream = lambda t: torch.zeros(99, dtype=torch.int64, device=DEV)
    with pytest.raises(BriefError):
        fb.run(2)
    fb.index_stream = lambda t: torch.zeros(100, dtype=torch.int32, device=DEV)
    with pytest.raises(BriefError):
        fb.run(2)


def test_in_kernel_sampling_equals_index_kernel():
    """the Philox stream drawn inside the fused kernel is the one brief_sample_indices writes"""
    dims = (24, 40, 56)
    pop = int(np.prod(dims))
    rng = np.random.default_rng(4)
    tv = torch.from_numpy(rng.uniform(0, 100, size=(pop, 1)).astype(np.float32)).to(DEV)
    ma, _, _ = make_net(4, 64, 20.0, seed=5)
    mb, _, _ = make_net(4, 64, 20.0, seed=5)
    n = 7001
    sa1, sa2, sb1, sb2 = (torch.zeros_like(ma.params) for _ in range(4))
    idx = torch.empty(n, dtype=torch.int64, device=DEV)
    for t in range(1, 4):
        _lib.check(_lib.lib().brief_sample_indices(_lib.ptr(idx), n, pop, 99, t, _lib.stream_ptr()))
        la = ma.fit_step(n, tv, 0, sa1, sa2, 1e-3, t, idx=idx, grid=(dims, -1.0, 1.0)).clone()
        lb = mb.fit_step(n, tv, 0, sb1, sb2, 1e-3, t, grid=(dims, -1.0, 1.0), rng=(pop, 99, t))
        assert torch.equal(la, lb) and torch.equal(ma.params, mb.params)


def test_gpu_ssim_vs_reference_golden(golden):
    from brief_pytorch_amd.metrics import gpu_ssim_u16
    g = golden("decode")
    a, b = torch.from_numpy(g["vol"].copy()).to(DEV), torch.from_numpy(g["dec_u16"].copy()).to(DEV)
    s, n = gpu_ssim_u16(a, b)
    assert n == g["vol"].shape[0] and abs(s / n - g["ssim"][0]) < 2e-5
    pa, pb = torch.from_numpy(g["pair_a"].copy()).to(DEV), torch.from_numpy(g["pair_b"].copy()).to(DEV)
    s, n = gpu_ssim_u16(pa, pb)
    assert abs(s / n - g["pair_ssim"][0]) < 2e-5
    # a larger, ragged volume against the oracle's float64 restatement
    from brief_pytorch_amd.synthetic import make_volume
    v = make_volume((7, 83, 141), seed=31)
    rng = np.random.default_rng(1)
    u = np.clip(v.astype(np.int64) + rng.integers(-400, 400, size=v.shape), 0, 65535).astype(np.uint16)
    s, n = gpu_ssim_u16(torch.from_numpy(v).to(DEV), torch.from_numpy(u).to(DEV))
    assert abs(s / n - O.ssim(v.astype(np.float32), u.astype(np.float32), 65535)) < 5e-5
