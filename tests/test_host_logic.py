"""Host-side mirror of the reference interface (normalise, weight files, metrics, schedules,
partition/budget/merge) against golden vectors produced by running the reference.  CPU only."""
import os

import numpy as np
import pytest
import torch

from brief_pytorch_amd import io as bio
from brief_pytorch_amd import metrics, misc, modelsave
from brief_pytorch_amd.networks import SIREN


def test_normalize_invnormalize_bit_exact(golden):
    g = golden("decode")
    d, side = bio.normalize_data(g["vol"], "minmaxany_0_100")
    assert np.array_equal(d.numpy(), g["norm_f32"])
    assert [side["min"], side["max"]] == list(g["side_min_max"])
    assert np.array_equal(bio.invnormalize_data(torch.from_numpy(g["inv_probe_in"].copy()), side, "minmaxany_0_100"), g["inv_probe_out"])
    assert np.array_equal(bio.invnormalize_data(torch.from_numpy(g["dec_f32"].copy()), side, "minmaxany_0_100"), g["dec_u16"])
    d8, side8 = bio.normalize_data(g["vol8"], "minmaxany_0_100")
    assert np.array_equal(d8.numpy(), g["norm8_f32"]) and side8["dtype"] == "uint8"
    assert np.array_equal(bio.invnormalize_data(d8.clone() * 0.97 + 1.0, side8, "minmaxany_0_100"), g["inv8"])


def test_metrics(golden):
    g = golden("decode")
    v32, o32 = g["vol"].astype(np.float32), g["dec_u16"].astype(np.float32)
    assert abs(metrics.cal_mse(v32, o32) - g["mse"][0]) / g["mse"][0] < 1e-6
    assert abs(metrics.cal_psnr(v32, o32, 65535) - g["psnr"][0]) < 1e-4
    assert abs(metrics.cal_ssim(v32, o32, 65535) - g["ssim"][0]) < 2e-5
    a, b = g["pair_a"].astype(np.float32), g["pair_b"].astype(np.float32)
    assert abs(metrics.cal_psnr(a, b, 65535) - g["pair_psnr"][0]) < 1e-4
    assert abs(metrics.cal_ssim(a, b, 65535) - g["pair_ssim"][0]) < 2e-5
    ia, ib = g["img_a"].astype(np.float32), g["img_b"].astype(np.float32)
    assert abs(metrics.cal_psnr(ia, ib, 255) - g["img_psnr"][0]) < 1e-4
    assert abs(metrics.cal_ssim(ia, ib, 255) - g["img_ssim"][0]) < 2e-5
    perf = metrics.eval_performance(7, g["vol"], g["dec_u16"], None, True, True, True)
    assert perf["steps"] == 7 and abs(perf["psnr"] - g["psnr"][0]) < 1e-4
    sse = float(((g["vol"].astype(np.int64) - g["dec_u16"].astype(np.int64)) ** 2).sum())
    assert abs(metrics.psnr_from_sse(sse, g["vol"].size, 65535) - g["psnr"][0]) < 1e-4


def test_model_files_roundtrip(tmp_path, golden):
    torch.manual_seed(3)
    m = SIREN(features=22, layers=5, w0=20)
    path = str(tmp_path / "module")
    modelsave.save_model(m, path)
    names = sorted(os.listdir(path))
    assert names == sorted(["weight-0-22-3", "bias-0-22", "weight-1-22-22", "bias-1-22", "weight-2-22-22", "bias-2-22",
                            "weight-3-22-22", "bias-3-22", "weight-4-1-22", "bias-4-1"])
    assert bio.get_folder_size(path) == 4 * m.param_count
    raw = np.fromfile(os.path.join(path, "weight-1-22-22"), dtype="<f4")    # row-major [out,in] float32
    assert np.array_equal(raw.reshape(22, 22), m.net[1][0].weight.data.numpy())
    torch.manual_seed(4)
    m2 = SIREN(features=22, layers=5, w0=20)
    modelsave.load_model(m2, path)
    assert torch.equal(m2.params, m.params)
    modelsave.CopyDir(path, str(tmp_path / "copy"))
    assert sorted(os.listdir(str(tmp_path / "copy"))) == names


def test_checkpoints_and_weights(golden):
    g = golden("divide")
    for i, key in enumerate(g["checkpoints_keys"]):
        spec, ms = str(key).split("@")
        assert misc.parse_checkpoints(spec, int(ms)) == list(g["checkpoints_%d" % i])
    for i, spec in enumerate((["value_65535_65535_1"], ["value_17000_20000_0.1"], ["quantile_16000_0.2_0.8_0.5"], ["exp_20000_0.5"], ["none"])):
        assert np.array_equal(misc.parse_weight(g["weight_vol"], spec), g["weight_%d" % i]), spec
    # the shipped default weight spec is all ones (prepare_fit then passes no weight map to the kernel)
    assert np.all(misc.parse_weight(g["vol"], ["value_65535_65535_1"]) == 1.0)
    assert not np.all(misc.parse_weight(g["vol"], ["value_17000_20000_0.1"]) == 1.0)


@pytest.mark.parametrize("dt", ["total_2_2_2", "every_5_8_7", "total_1_2_4"])
def test_divide_alloc_merge(golden, dt):
    g = golden("divide")
    vol = g["vol"]
    chunks, outline = misc.divide_data(vol, dt)
    assert [c["name"] for c in chunks] == list(g["names_" + dt])
    assert [c["size"] for c in chunks] == list(g["sizes_" + dt])
    assert outline.shape == vol.shape and (outline == 2000).any()
    for alloc in ("equal", "by_size", "by_var", "by_d", "by_dv"):
        import copy
        ch = misc.alloc_param(copy.deepcopy(chunks), 40000.0, alloc, 26)
        assert [c["name"] for c in ch] == list(g["alloc_%s_%s_names" % (dt, alloc)])
        np.testing.assert_allclose([c["param_size"] for c in ch], g["alloc_%s_%s" % (dt, alloc)], rtol=1e-12)
    dl = [{"data": c["data"].copy(), "name": c["name"], **misc.parse_chunk_name(c["name"])} for c in chunks]
    assert np.array_equal(misc.merge_divided_data(dl, list(vol.shape)), g["merge_" + dt])
    assert np.array_equal(g["merge_" + dt], vol)


def test_budget_drop_and_divide_num(golden):
    g = golden("divide")
    import copy
    chunks, _ = misc.divide_data(g["vol"], "every_11_16_20")
    ch = misc.alloc_param(copy.deepcopy(chunks), 300.0, "by_size", 26)
    assert [c["name"] for c in ch] == list(g["drop_names"])
    np.testing.assert_allclose([c["param_size"] for c in ch], g["drop_sizes"], rtol=1e-12)
    assert abs(misc.cal_feature(g["vol"]) - g["feature_vol"][0]) < 1e-15
    for d, h, w, nb, nd, nh, nw in g["divide_num"]:
        assert misc.cal_divide_num(int(d), int(h), int(w), int(nb), 1e6) == (nd, nh, nw)
    assert misc.cal_divide_num(64, 64, 64, -1, 4 * 1361 * 9.5) == tuple(g["divide_num_auto"])


def test_coordinate_grids_match_the_reference(golden):
    """create_coords / create_flattened_coords (utils/dataset.py:11-62): the axis values are the reference's
    torch.linspace bit for bit (golden linspace_*), ordering is 'ij' with the last axis fastest"""
    import torch
    from brief_pytorch_amd.dataset import create_coords, create_flattened_coords
    g = golden("decode")
    for n in (2, 3, 16, 63, 64):
        c = create_coords((n, 5), "-1,1")
        assert c.shape == (n, 5, 2) and np.array_equal(c[:, 0, 0].numpy(), g["linspace_%d" % n])
    f = create_flattened_coords((3, 4, 5), "n11")
    assert f.shape == (60, 3)
    assert np.array_equal(f.numpy(), g["flat_coords_3_4_5"])                       # the reference's own output
    assert np.array_equal(create_flattened_coords((4, 6), "-1,1").numpy(), g["flat_coords_4_6"])
    assert torch.equal(f[7], torch.tensor([-1.0, torch.linspace(-1, 1, 4)[1], torch.linspace(-1, 1, 5)[2]]))
    assert torch.equal(create_flattened_coords((4, 5), "0p1")[6], torch.tensor([torch.linspace(0, 1, 4)[1], torch.linspace(0, 1, 5)[1]]))
    with pytest.raises(NotImplementedError):
        create_coords((2, 2, 2, 2))


def test_reconstruct_flattened_chunks():
    import torch
    from brief_pytorch_amd.dataset import create_flattened_coords, reconstruct_flattened
    calls = []

    def nf(c):
        calls.append(c.shape[0])
        return c.sum(-1, keepdim=True)
    out = reconstruct_flattened((3, 4, 5, 1), 16, nf, coords_mode="-1,1")
    assert calls == [16, 16, 16, 12] and out.shape == (3, 4, 5, 1)
    assert torch.equal(out.reshape(-1, 1), create_flattened_coords((3, 4, 5), "-1,1").sum(-1, keepdim=True))
    # half=True (utils/misc.py:70-71, 78-79, 85-86): fp16 coordinates go in, an fp16 volume comes out
    seen = []

    def nf16(c):
        seen.append(c.dtype)
        return c.float().sum(-1, keepdim=True)
    out16 = reconstruct_flattened((3, 4, 5, 1), 16, nf16, half=True, coords_mode="-1,1")
    assert out16.dtype == torch.float16 and set(seen) == {torch.float16}
    assert torch.equal(out16.reshape(-1, 1), create_flattened_coords((3, 4, 5), "-1,1").half().float().sum(-1, keepdim=True).half())


def test_configure_optimizer_and_scheduler_mirror():
    import torch
    p = [torch.zeros(3, requires_grad=True)]
    for name, cls in (("Adam", torch.optim.Adam), ("Adamax", torch.optim.Adamax), ("SGD", torch.optim.SGD)):
        o = misc.configure_optimizer(p, name, 1e-3)
        assert isinstance(o, cls) and o.param_groups[0]["lr"] == 1e-3
    with pytest.raises(NotImplementedError):
        misc.configure_optimizer(p, "LBFGS", 1.0)
    o = misc.configure_optimizer(p, "Adamax", 1e-3)
    s = misc.configure_lr_scheduler(o, {"name": "MultiStepLR", "milestones": [2, 4], "gamma": 0.2})
    lrs = []
    for _ in range(5):
        lrs.append(o.param_groups[0]["lr"]); o.step(); s.step()
    assert np.allclose(lrs, [1e-3, 1e-3, 2e-4, 2e-4, 4e-5])
    assert isinstance(misc.configure_lr_scheduler(o, {"name": "none"}), torch.optim.lr_scheduler.MultiStepLR)
    assert isinstance(misc.configure_lr_scheduler(o, {"name": "StepLR", "step_size": 3}), torch.optim.lr_scheduler.StepLR)
    with pytest.raises(NotImplementedError):
        misc.configure_lr_scheduler(o, {"name": "Cosine"})


def test_windowed_cube_sampler_index_stream_matches_the_reference(golden):
    """RandomCubeSampler with windows smaller than the volume (main.py:38-125): same window draws (global torch RNG after
    the net's init draws), same window enumeration, same voxel order inside and across windows"""
    from brief_pytorch_amd.framework import _CubeIndexStream
    g = golden("cube")
    L, F, w0, steps, count = (int(v) for v in g["cfg"][:5])
    cl = [int(v) for v in g["cfg"][5:]]
    dims = g["vol"].shape[:3]
    torch.manual_seed(42)                                   # reproduc(opt.Reproduc), main.py:653-661
    m = SIREN(features=F, layers=L, w0=w0)                  # prepare_module consumes the generator first
    for l in range(L):
        assert np.array_equal(m.net[l][0].weight.data.numpy(), g["init_w%d" % l])
    st = _CubeIndexStream(dims, cl, count, "cpu")
    assert st.pop_size == int(g["pop_size"][0]) and st.n == count * int(np.prod(cl))
    for t in range(steps):
        assert np.array_equal(st(t + 1).numpy(), g["voxels"][t]), t
    # an explicit generator keeps the stream away from the global one (tests that must not disturb other draws)
    gen = torch.Generator().manual_seed(1)
    a = _CubeIndexStream(dims, cl, count, "cpu", generator=gen)(1)
    assert a.numel() == st.n and int(a.max()) < int(np.prod(dims))


def test_torch_rng_point_sampler_reproduces_the_reference_stream_from_the_seed_alone(golden):
    """Compress.sampler.rng: torch — RandompointSampler's own draws (main.py:154-163: torch.randint(0, pop, (sample_size,)) on the CPU
    generator, once per step, behind reproduc(seed) and the net's init draws).  tests/golden/trace.npz recorded the reference's index
    sets of 50 steps (seed 42, 4 x 32 net, 12 x 20 x 28 volume, 1 000 samples per step): the stream must BE them, draw for draw,
    whether it is asked step by step or for runs of steps at once."""
    from brief_pytorch_amd.framework import _PointIndexStream
    g = golden("trace")
    pop = int(np.prod(g["pt_vol"].shape[:3]))
    for how in ("steps", "runs"):
        torch.manual_seed(42)                               # reproduc(opt.Reproduc), main.py:653-661
        m = SIREN(features=32, layers=4, w0=20)             # prepare_module consumes the generator first
        assert np.array_equal(m.net[0][0].weight.data.numpy(), g["pt_init_w0"])
        gen = torch.Generator()
        gen.set_state(torch.get_rng_state())                # NFGR.prepare_fit's private fork
        st = _PointIndexStream(pop, 1000, "cpu", generator=gen)
        if how == "steps":
            got = np.stack([st(t + 1).numpy() for t in range(50)])
        else:
            got = np.concatenate([st.batch(1, 7).numpy(), st.batch(8, 43).numpy()])
        assert np.array_equal(got, g["pt_idx"]), how


@pytest.mark.parametrize("kw", [dict(base_lr=1e-4, max_lr=1e-3, step_size_up=7), dict(base_lr=1e-4, max_lr=2e-3, step_size_up=5, step_size_down=9, mode="triangular2"),
                                dict(base_lr=2e-4, max_lr=1e-3, step_size_up=4, mode="exp_range", gamma=0.97),
                                dict(base_lr=1e-4, max_lr=1e-3, step_size_up=6, cycle_momentum=False)])
def test_cyclic_lr_matches_torch(kw):
    """CyclicLR (utils/misc.py:188-189 passes the YAML keys to torch): lr and, for Adam-family optimizers, beta1 of every
    optimizer step against torch's own scheduler"""
    from brief_pytorch_amd.fit import cyclic_lr
    p = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.Adamax([p], lr=1e-3)
    sch = torch.optim.lr_scheduler.CyclicLR(opt, **kw)
    lr_at, b1_at = cyclic_lr(**kw)
    for t in range(1, 60):
        assert abs(opt.param_groups[0]["lr"] - lr_at(t)) <= 1e-12 * abs(lr_at(t)), t
        if kw.get("cycle_momentum", True):
            assert abs(opt.param_groups[0]["betas"][0] - b1_at(t)) <= 1e-12, t
        else:
            assert b1_at(t) is None and opt.param_groups[0]["betas"][0] == 0.9
        p.grad = torch.ones(3)
        opt.step()
        sch.step()


def test_multitask_expansion(tmp_path):
    """MultiTask.py:27-85: PRODUCT / CONCAT expansion of dotted overrides over the Static tree, one YAML per combination"""
    import importlib.util
    import yaml
    spec = importlib.util.spec_from_file_location("MultiTask", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "MultiTask.py"))
    mt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mt)
    dyn = [{"PRODUCT": [{"CONCAT": [{"a.x": 1, "a.y": "p"}, {"a.x": 2}]},
                        {"CONCAT": [{"b.k": "u", "Log.project_name": "U"}, {"b.k": "v", "Log.project_name": "V"}, {"b.k": "w"}]}]},
           {"a.x": 9}]
    combos = mt.expand({"CONCAT": dyn})
    assert len(combos) == 2 * 3 + 1
    assert combos[0] == [("a.x", 1), ("a.y", "p"), ("b.k", "u"), ("Log.project_name", "U")] and combos[-1] == [("a.x", 9)]
    sweep = {"Dynamic": dyn, "Static": {"Source": {"gpucost": 1}, "Log": {"project_name": "base", "time": True}, "a": {"x": 0, "z": [1, 2]}, "b": {"k": "none"}}}
    path = tmp_path / "sweep.yaml"
    path.write_text(yaml.safe_dump(sweep))
    tasks, temp_dir = mt.gen_task_list(str(path), "main.py")
    assert [t[0] for t in tasks] == ["exp_%03d" % i for i in range(7)] and os.path.basename(temp_dir) == "temp_opt_base"
    t1 = yaml.safe_load(open(tasks[1][2]))
    assert t1 == {"Log": {"project_name": "V", "time": True}, "a": {"x": 1, "z": [1, 2], "y": "p"}, "b": {"k": "v"}}
    assert tasks[1][1][-2:] == ["-p", tasks[1][2]]
    # the shipped sweep file expands and every task is a loadable SingleTask option tree
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tasks, temp_dir = mt.gen_task_list(os.path.join(root, "opt", "MultiTask", "default.yaml"), "main.py")
    try:
        assert len(tasks) == 2
        kinds = sorted(yaml.safe_load(open(t[2]))["CompressFramework"]["Compress"]["divide"]["divide_type"] for t in tasks)
        assert kinds == ["adaptive_-1_-1_0_0_20", "none"]
    finally:
        import shutil
        shutil.rmtree(temp_dir, ignore_errors=True)
