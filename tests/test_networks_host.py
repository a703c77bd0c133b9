"""Host-side pieces of the SIREN drop-in against goldens from the reference (CPU only)."""
import ctypes as C

import numpy as np
import pytest
import torch

from brief_pytorch_amd import _lib
from brief_pytorch_amd.networks import SIREN, get_nnmodule_param_count, init_phi


@pytest.mark.parametrize("seed", [0, 42])
@pytest.mark.parametrize("LF", [(3, 64), (5, 22), (7, 56)])
def test_init_replays_reference_rng(golden, seed, LF):
    g = golden("init")
    L, F = LF
    torch.manual_seed(seed)
    m = SIREN(coords_channel=3, data_channel=1, features=F, layers=L, w0=20)
    for l in range(L):
        assert np.array_equal(m.net[l][0].weight.data.numpy(), g["s%d_L%d_F%d_w%d" % (seed, L, F, l)])
        assert np.array_equal(m.net[l][0].bias.data.numpy(), g["s%d_L%d_F%d_b%d" % (seed, L, F, l)])


@pytest.mark.parametrize("seed", [0, 42])
def test_init_256(golden, seed):
    g = golden("init")
    torch.manual_seed(seed)
    m = SIREN(features=256, layers=5, w0=20)
    sd = {("w%d" % l): m.net[l][0].weight.data.numpy() for l in range(5)}
    sd.update({("b%d" % l): m.net[l][0].bias.data.numpy() for l in range(5)})
    sums = np.array([sd[k].astype(np.float64).sum() for k in sorted(sd)])
    assert np.array_equal(sums, g["s%d_L5_F256_sums" % seed])
    assert np.array_equal(sd["w1"][0], g["s%d_L5_F256_w1_row0" % seed])
    assert np.array_equal(sd["b3"], g["s%d_L5_F256_b3" % seed])


def test_budget_table(golden):
    g = golden("budget")
    for L, cin, cout, nbytes, F, P in g["table"]:
        f = SIREN.calc_features(nbytes / 4.0, int(cin), int(cout), int(L), False)
        assert f == int(F)
        assert SIREN.calc_param_count(int(cin), int(cout), f, int(L), False) == int(P)


def test_module_surface():
    torch.manual_seed(1)
    m = init_phi({"name": "SIREN", "coords_channel": 3, "data_channel": 1, "layers": 5, "w0": 20,
                  "output_act": False, "res": False, "features": 22})
    assert get_nnmodule_param_count(m) == SIREN.calc_param_count(3, 1, 22, 5) == m.params.numel()
    assert len(m.net) == 5 and m.net[1][0].weight.shape == (22, 22)
    w = torch.arange(22 * 22, dtype=torch.float32).reshape(22, 22)
    m.net[2][0].weight.data = w                      # utils/ModelSave.py:20 style assignment
    assert torch.equal(m.net[2][0].weight.data, w) and m._stale
    assert torch.equal(m.state_dict()["net.2.0.weight"], w)
    with pytest.raises(NotImplementedError):
        init_phi({"name": "NeRF"})
    with pytest.raises(NotImplementedError):
        SIREN(res=True)
    with pytest.raises(_lib.BriefError):           # no CPU fallback
        m.forward(torch.zeros(4, 3))
    # .half() / .float() (main.py:212, 287-288, 389, 398): the low-precision mode and back; parameters stay fp32 masters
    p0 = m.params.clone()
    assert m.half() is m and m.precision == "bf16" and m.desc.precision == _lib.PRECISION["bf16"] and m.params.dtype == torch.float32
    assert m.float() is m and m.precision == "fp32" and m.desc.precision == 0 and torch.equal(m.params, p0)
    wide = SIREN(features=600, layers=3)
    assert wide.half().precision == "fp32"          # no bf16 kernels above 512 features: stays exact


def test_capi_exports_and_sizes():
    import os
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libbrief_hip.so not built")
    L = _lib.lib()
    for name in _lib.EXPORTS:
        assert hasattr(L, name), name
    assert L.brief_version() == 130
    d = _lib.SirenDesc(3, 1, 5, 256, 20.0, 30.0, 0, 0)
    assert L.brief_param_count(C.byref(d)) == 198657
    assert L.brief_packed_count(C.byref(d)) == 256 * 4 + 3 * (2 * 256 * 256 + 256) + 4 * 256 + 4
    assert L.brief_train_workspace_bytes(C.byref(d), 100000) > 6 * 256 * 100000 * 4
    wide = _lib.SirenDesc(3, 1, 9, 512, 20.0, 30.0, 0, 0)
    assert L.brief_param_count(C.byref(wide)) == 1841153          # BASELINE config 3 network (8x512)
    assert L.brief_packed_count(C.byref(_lib.SirenDesc(3, 1, 5, 300, 20.0, 30.0, 0, 0))) == 320 * 4 + 3 * (2 * 320 * 320 + 320) + 4 * 320 + 4      # whole 32-feature tiles (round 3: 384)
    # above 512 features the tile count is exact (k_lean walks a run-time number of tiles): the shipped default.yaml on a 512^3
    # uint16 volume solves to F = 527 = 17 tiles (utils/Networks.py:299-314 has no width limit)
    w527 = _lib.SirenDesc(3, 1, 5, 527, 20.0, 30.0, 0, 0)
    assert L.brief_param_count(C.byref(w527)) == 527 * 3 + 527 + 3 * (527 * 527 + 527) + 527 + 1
    assert L.brief_packed_count(C.byref(w527)) == 544 * 4 + 3 * (2 * 544 * 544 + 544) + 4 * 544 + 4
    assert L.brief_packed_count(C.byref(_lib.SirenDesc(3, 1, 5, 1024, 20.0, 30.0, 0, 0))) == 1024 * 4 + 3 * (2 * 1024 * 1024 + 1024) + 4 * 1024 + 4
    assert L.brief_train_workspace_bytes(C.byref(w527), 100000) > 6 * 544 * 100000 * 4
    # above 1024 features (k_wide): still exact tiles; the default.yaml budget of a 1024^3 uint16 volume solves to 1494 = 47 tiles.  Inference
    # needs a scratch there (two ping-pong planes per workgroup), training one more phase plane than narrower nets
    w1494 = _lib.SirenDesc(3, 1, 5, 1494, 20.0, 30.0, 0, 0)
    assert L.brief_packed_count(C.byref(w1494)) == 1504 * 4 + 3 * (2 * 1504 * 1504 + 1504) + 4 * 1504 + 4
    assert L.brief_forward_workspace_bytes(C.byref(w1494), 32 * 7) == 7 * 2 * 1504 * 32 * 4
    assert L.brief_forward_workspace_bytes(C.byref(w527), 100000) == 0
    assert L.brief_train_workspace_bytes(C.byref(w1494), 100000) > 7 * 1504 * 100000 * 4
    bad = _lib.SirenDesc(3, 1, 5, 4097, 20.0, 30.0, 0, 0)
    assert L.brief_param_count(C.byref(bad)) == -1 and b"features" in L.brief_last_error()
    assert L.brief_param_count(C.byref(_lib.SirenDesc(3, 1, 5, 513, 20.0, 30.0, 0, 1))) == -1 and b"BF16" in L.brief_last_error()


def test_every_function_the_header_declares_is_exported():
    """include/brief_hip.h is the contract: each `brief_*(` prototype must be a symbol of libbrief_hip.so and be
    listed in the ctypes layer (no compute calls: runs without a GPU)"""
    import os
    import re
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libbrief_hip.so not built")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "brief_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)                      # prototypes only, not prose
    names = sorted(set(re.findall(r"\b(brief_[a-z0-9_]+)\s*\(", text)))
    assert len(names) >= 19 and "brief_siren_fit_step" in names and "brief_multi_fit" in names
    L = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "declared in brief_hip.h but not exported: " + n
        assert n in _lib.EXPORTS, "exported but missing from brief_pytorch_amd/_lib.py: " + n
    assert sorted(_lib.EXPORTS) == names


def test_product_library_reads_no_environment_variable():
    """Result- or launch-changing diagnostics (BRIEF_DIAG, BRIEF_TAIL_ROUNDS, BRIEF_WGRAD_REPEAT, ...) exist only in a
    -DBRIEF_DIAGNOSTICS build: the shipped library must not import getenv at all."""
    import os
    import subprocess
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libbrief_hip.so not built")
    if os.path.basename(_lib.LIB_PATH) != "libbrief_hip.so":
        pytest.skip("BRIEF_LIB points at a diagnostic build")
    syms = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "hipLaunchKernel" in syms or "hipModuleLaunchKernel" in syms      # the listing is what we think it is
    assert "getenv" not in syms, "the product libbrief_hip.so imports getenv"
    src = open(_lib.SRC).read() + open(os.path.join(os.path.dirname(_lib.SRC), "brief_bf16.inc")).read()
    assert src.count("getenv(") == 1 and "#ifdef BRIEF_DIAGNOSTICS\nstatic const char *env_str(const char *name) { return getenv(name); }" in src


def test_bf16_packed_layout_counts():
    """BRIEF_PREC_BF16: widths pad to 256 / 512 and the packed buffer grows by the bf16 W / W^T fragment copies
    (host-side layout arithmetic only: no GPU needed)"""
    import ctypes as C
    L = _lib.lib()
    d = _lib.SirenDesc(3, 1, 9, 512, 20.0, 30.0, 0, 1)
    fp = 512
    c32 = fp * 4 + 7 * (2 * fp * fp + fp) + 4 * fp + 4
    assert L.brief_packed_count(C.byref(d)) == (c32 + 3) // 4 * 4 + 7 * fp * fp
    d2 = _lib.SirenDesc(3, 1, 5, 40, 20.0, 30.0, 0, 1)
    assert L.brief_packed_count(C.byref(d2)) == (256 * 4 + 3 * (2 * 256 * 256 + 256) + 4 * 256 + 4 + 3) // 4 * 4 + 3 * 256 * 256
    assert L.brief_param_count(C.byref(d2)) == 40 * 3 + 40 + 3 * (40 * 40 + 40) + 40 + 1
    plane = 8 * 512 * 100096 * 2                     # one stash plane: 8 sine layers x 512 features x padded samples x 2 bytes
    assert 2 * plane < L.brief_train_workspace_bytes(C.byref(d), 100000) < 2.2 * plane      # the fp16 phase planes + the bf16 delta planes (round 3: no cosine planes)
    bad = _lib.SirenDesc(3, 1, 5, 256, 20.0, 30.0, 0, 7)
    assert L.brief_packed_count(C.byref(bad)) < 0 and b"precision" in L.brief_last_error()


def test_bf16x3_packed_layout_counts_and_limits():
    """BRIEF_PREC_BF16X3: every net runs on the 256-wide tile; the packed buffer is the f32 buffer + one hi and one lo bf16
    fragment region (W and W^T of each hidden layer: FP^2 dwords per layer and half); widths above 256 are refused"""
    import ctypes as C
    L = _lib.lib()
    assert _lib.PRECISION["bf16x3"] == 2
    c32 = 256 * 4 + 3 * (2 * 256 * 256 + 256) + 4 * 256 + 4
    for F in (256, 200, 22):
        d = _lib.SirenDesc(3, 1, 5, F, 20.0, 30.0, 0, 2)
        assert L.brief_packed_count(C.byref(d)) == (c32 + 3) // 4 * 4 + 2 * 3 * 256 * 256
    wide = _lib.SirenDesc(3, 1, 5, 300, 20.0, 30.0, 0, 2)
    assert L.brief_packed_count(C.byref(wide)) < 0 and b"256" in L.brief_last_error()
