"""Pins the CPU oracle (oracle/) against golden vectors produced by RUNNING the reference
(tests/golden/make_golden.py).  CPU-only; part of `-m "not gpu"`."""
import numpy as np
import pytest

from oracle import oracle as O


def _net(g, tag, L):
    ws = [g["%s_w%d" % (tag, l)] for l in range(L)]
    bs = [g["%s_b%d" % (tag, l)] for l in range(L)]
    return ws, bs


def relerr(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))) / (np.max(np.abs(b)) + 1e-30))


@pytest.mark.parametrize("n", [2, 3, 16, 63, 64, 100, 512])
def test_linspace_bit_exact(golden, n):
    g = golden("decode")
    assert np.array_equal(O.linspace(-1.0, 1.0, n), g["linspace_%d" % n])


def test_linspace01_and_grid(golden):
    g = golden("decode")
    assert np.array_equal(O.linspace(0.0, 1.0, 37), g["linspace01_37"])
    assert np.array_equal(O.grid_coords((3, 4, 5)), g["flat_coords_3_4_5"])
    assert np.array_equal(O.grid_coords((4, 6)), g["flat_coords_4_6"])
    idx = np.array([0, 59, 17, 17, 33], np.int64)
    assert np.array_equal(O.grid_coords((3, 4, 5), idx=idx), g["flat_coords_3_4_5"][idx])


@pytest.mark.parametrize("tag", ["a", "b", "c", "e", "f"])
def test_forward(golden, tag):
    g = golden("forward")
    L, F, w0, cin, cout, oa = [int(v) for v in g[tag + "_cfg"]]
    d = O.make_desc(cin, cout, L, F, w0, 30.0, bool(oa))
    p = O.pack_params(*_net(g, tag, L))
    assert p.size == O.param_count(d)
    y = O.forward(d, p, g[tag + "_x"])
    y64 = O.forward(d, p, g[tag + "_x"], f64=True)
    # reference fp32 (MKL sgemm + SLEEF sin) vs restatement: rounding-level agreement
    assert relerr(y, g[tag + "_y"]) < 2e-5
    assert relerr(y64, g[tag + "_y"]) < 2e-5


@pytest.mark.parametrize("tag", ["mse_unit", "mse_unit_thr", "mse_w_thr", "sl1_w"])
def test_loss_grads(golden, tag):
    g = golden("grads")
    L, F, w0, kind, thr, beta = g[tag + "_cfg"]
    L, F, kind = int(L), int(F), int(kind)
    d = O.make_desc(3, 1, L, F, w0)
    p = O.pack_params(*_net(g, tag, L))
    loss, grads, yh, we = O.loss_grad(d, p, g[tag + "_x"], g[tag + "_y"], g[tag + "_w"], kind, float(thr), float(beta))
    assert abs(loss - g[tag + "_loss"][0]) / abs(g[tag + "_loss"][0]) < 2e-6
    assert relerr(yh, g[tag + "_yhat"]) < 2e-5
    assert np.array_equal(we, g[tag + "_w_after"])
    gw, gb = O.unpack_params(d, grads)
    for l in range(L):
        assert relerr(gw[l], g["%s_gw%d" % (tag, l)]) < 1e-4, l
        assert relerr(gb[l], g["%s_gb%d" % (tag, l)]) < 1e-4, l
    # fp64 restatement agrees too (logic check independent of f32 rounding)
    loss64, grads64, _, _ = O.loss_grad(d, p, g[tag + "_x"], g[tag + "_y"], g[tag + "_w"], kind, float(thr), float(beta), f64=True)
    assert relerr(grads64, grads) < 1e-4


@pytest.mark.parametrize("name", ["Adamax", "Adam", "SGD"])
def test_optimizers(golden, name):
    g = golden("optim")
    L = 4
    d = O.make_desc(3, 1, L, 24, 20.0)
    p = O.pack_params([g["init_w%d" % l] for l in range(L)], [g["init_b%d" % l] for l in range(L)])
    s1, s2 = np.zeros_like(p), np.zeros_like(p)
    lrs = O.multistep_lr(1e-3, [3, 6], 0.2, 10)
    # SGD at lr 1e-3 on this loss takes O(1) weight steps through sin(30 z): chaotic after ~3 steps,
    # so only the first steps pin the update rule (the reference diverges from itself there too)
    last = 3 if name == "SGD" else 10
    for t in range(1, last + 1):
        loss, grads, _, _ = O.loss_grad(d, p, g["x"], g["y"])
        assert abs(loss - g["%s_loss_t%d" % (name, t)][0]) / g["%s_loss_t%d" % (name, t)][0] < 5e-6, t
        O.optim_step(name, p, grads, s1, s2, lrs[t - 1], t)
        if t in (1, 2, 10):
            ws, bs = O.unpack_params(d, p)
            for l in range(L):
                assert np.max(np.abs(ws[l] - g["%s_t%d_w%d" % (name, t, l)])) < 2e-6, (t, l)
                assert np.max(np.abs(bs[l] - g["%s_t%d_b%d" % (name, t, l)])) < 2e-6, (t, l)
            if name != "SGD":
                m_w, m_b = O.unpack_params(d, s1)
                u_w, u_b = O.unpack_params(d, s2)
                for l in range(L):
                    assert relerr(m_w[l], g["%s_t%d_p%d_s1" % (name, t, 2 * l)]) < 1e-4
                    assert relerr(u_w[l], g["%s_t%d_p%d_s2" % (name, t, 2 * l)]) < 1e-4
                    assert relerr(m_b[l], g["%s_t%d_p%d_s1" % (name, t, 2 * l + 1)]) < 1e-4
                    assert relerr(u_b[l], g["%s_t%d_p%d_s2" % (name, t, 2 * l + 1)]) < 1e-4


def test_optim_update_rule_exact(golden):
    """Given the reference's own gradient (recovered from its Adamax state at t=1:
    exp_avg = 0.1*g), one oracle update reproduces the reference parameters to 1 ulp."""
    g = golden("optim")
    for name in ("Adamax", "Adam"):
        for pi, key in enumerate(["w0", "b0", "w1", "b1"]):
            p0 = g["init_" + key].ravel().copy()
            m1 = g["%s_t1_p%d_s1" % (name, pi)].ravel()
            grad = (m1.astype(np.float64) / (1.0 - 0.9)).astype(np.float32)
            s1, s2 = np.zeros_like(p0), np.zeros_like(p0)
            O.optim_step(name, p0, grad, s1, s2, 1e-3, 1)
            ref = g["%s_t1_%s" % (name, key)].ravel()
            assert np.max(np.abs(p0 - ref)) <= 2 * np.spacing(np.abs(ref).max()), (name, key)


def test_trace_full_batch(golden):
    g = golden("trace")
    d = O.make_desc(3, 1, 5, 22, 20.0)
    p = O.pack_params([g["cube_init_w%d" % l] for l in range(5)], [g["cube_init_b%d" % l] for l in range(5)])
    vol = g["cube_vol"]
    vn, side = O.normalize(vol)
    thr = float(g["cube_thr"][0])
    pf, losses, _, _ = O.fit(d, p, vn.reshape(-1, 1), vol.shape[:3], 50, thr=thr)
    ref = g["cube_losses"]
    assert np.max(np.abs(losses - ref) / ref) < 1e-4
    wf, bf = O.unpack_params(d, pf)
    for l in range(5):
        assert np.max(np.abs(wf[l] - g["cube_final_w%d" % l])) < 5e-5
        assert np.max(np.abs(bf[l] - g["cube_final_b%d" % l])) < 5e-5


def test_trace_randompoint(golden):
    g = golden("trace")
    d = O.make_desc(3, 1, 4, 32, 20.0)
    p = O.pack_params([g["pt_init_w%d" % l] for l in range(4)], [g["pt_init_b%d" % l] for l in range(4)])
    vol = g["pt_vol"]
    vn, side = O.normalize(vol)
    thr = float(O.normalize(np.array([65535], np.uint16), vmin=side["min"], vmax=side["max"])[0][0])
    pf, losses, _, _ = O.fit(d, p, vn.reshape(-1, 1), vol.shape[:3], 50, idx_stream=g["pt_idx"], thr=thr)
    ref = g["pt_losses"]
    assert np.max(np.abs(losses - ref) / ref) < 1e-4
    wf, bf = O.unpack_params(d, pf)
    for l in range(4):
        assert np.max(np.abs(wf[l] - g["pt_final_w%d" % l])) < 5e-5


def test_normalize_invnormalize_bit_exact(golden):
    g = golden("decode")
    vn, side = O.normalize(g["vol"])
    assert np.array_equal(vn, g["norm_f32"])
    assert side["min"] == g["side_min_max"][0] and side["max"] == g["side_min_max"][1]
    assert np.array_equal(O.invnormalize(g["inv_probe_in"], side), g["inv_probe_out"])
    assert np.array_equal(O.invnormalize(g["dec_f32"], side), g["dec_u16"])
    vn8, side8 = O.normalize(g["vol8"])
    assert np.array_equal(vn8, g["norm8_f32"])
    assert np.array_equal(O.invnormalize(g["norm8_f32"] * np.float32(0.97) + np.float32(1.0), side8), g["inv8"])


def test_decode_and_metrics(golden):
    g = golden("decode")
    d = O.make_desc(3, 1, 5, 22, 20.0)
    p = O.pack_params([g["net_w%d" % l] for l in range(5)], [g["net_b%d" % l] for l in range(5)])
    vol = g["vol"]
    dec = O.decode(d, p, vol.shape[:3])
    assert relerr(dec, g["dec_f32"]) < 2e-5
    _, side = O.normalize(vol)
    u16 = O.invnormalize(dec, side)
    diff = np.abs(u16.astype(np.int64) - g["dec_u16"].astype(np.int64))
    assert diff.max() <= 1 and (diff != 0).mean() < 0.02      # truncation may flip a count on rounding-level yhat noise
    assert abs(O.psnr(vol, g["dec_u16"], 65535) - g["psnr"][0]) < 1e-4
    assert abs(O.ssim(vol.astype(np.float32), g["dec_u16"].astype(np.float32), 65535) - g["ssim"][0]) < 1e-5
    assert abs(O.psnr(g["pair_a"], g["pair_b"], 65535) - g["pair_psnr"][0]) < 1e-4
    assert abs(O.ssim(g["pair_a"].astype(np.float32), g["pair_b"].astype(np.float32), 65535) - g["pair_ssim"][0]) < 1e-5
    assert abs(O.psnr(g["img_a"], g["img_b"], 255) - g["img_psnr"][0]) < 1e-4
    assert abs(O.ssim(g["img_a"].astype(np.float32), g["img_b"].astype(np.float32), 255) - g["img_ssim"][0]) < 1e-5


def test_deblock_oracle_vs_reference_python(golden):
    g = golden("deblock")
    names = [str(n) for n in g["names"]]
    assert O.deblock_lines(names) == g["lines"].tolist()
    assert np.array_equal(O.deblock(g["img"], names), g["out"])
    assert np.array_equal(O.deblock(g["img"], names, 48, 700, 20050), g["out2"])
    # the C++ flavour (integer arithmetic) is a different filter: it must differ somewhere, not everywhere
    oc = O.deblock(g["img"], names, mode=0)
    assert 0 < (oc != g["out"]).sum() < 0.2 * (g["img"] != g["out"]).sum()
