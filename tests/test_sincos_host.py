"""Accuracy of the kernels' sine / cosine (csrc/brief_math.h) against float64.

Sine.forward of the reference (utils/Networks.py:227-234) goes through torch's ~1-ulp sin; the kernels use an exact
two-term reduction to revolutions followed by v_sin_f32 / v_cos_f32.  The host build of the header (g++) measures the
REDUCTION (the hardware ops are replaced by float64 sin/cos of the reduced argument there) and the software
brief_sincosf; the GPU test measures the whole device path through brief_sincos_probe.  Bound stated in brief_math.h:
total |err| <= ~3.1e-7 for the |w0 z| <= ~200 rad a SIREN produces."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include <math.h>
#include "brief_math.h"
extern "C" {
void probe_fast(const float *x, float *s, float *c, float *rev, long n) { for (long i = 0; i < n; ++i) { brief_fast_sincosf(x[i], s + i, c + i); rev[i] = brief_revolutions(x[i]); } }
void probe_soft(const float *x, float *s, float *c, long n) { for (long i = 0; i < n; ++i) brief_sincosf(x[i], s + i, c + i); }
}
'''


@pytest.fixture(scope="module")
def hostlib(tmp_path_factory):
    d = tmp_path_factory.mktemp("sincos")
    src, so = str(d / "probe.cpp"), str(d / "probe.so")
    with open(src, "w") as f:
        f.write(SRC)
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(ROOT, "brief_pytorch_amd", "csrc"), src, "-o", so])
    return C.CDLL(so)


def _inputs():
    rng = np.random.default_rng(0)
    return np.concatenate([rng.uniform(-200, 200, 400000), rng.uniform(-35, 35, 400000), np.linspace(-3.2, 3.2, 20001),
                           np.arange(-63, 64) * (np.pi / 2), [0.0, 1e-30, -1e-30]]).astype(np.float32)


def test_reduction_to_revolutions_and_software_sincos(hostlib):
    x = _inputs()
    n = x.size
    fp = C.POINTER(C.c_float)
    s, c, rev = np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, np.float32)
    hostlib.probe_fast(x.ctypes.data_as(fp), s.ctypes.data_as(fp), c.ctypes.data_as(fp), rev.ctypes.data_as(fp), C.c_long(n))
    x64 = x.astype(np.float64)
    turns = x64 / (2 * np.pi)
    err_rad = np.abs((rev.astype(np.float64) - (turns - np.rint(turns))) * 2 * np.pi)
    err_rad = np.minimum(err_rad, np.abs(err_rad - 2 * np.pi))            # rint ties: +-0.5 revolutions are the same angle
    assert np.abs(rev).max() <= 0.5 + 1e-6
    assert err_rad.max() < 1.9e-7, err_rad.max()                          # argument error of the two-term reduction
    assert np.abs(s - np.sin(x64)).max() < 2.6e-7 and np.abs(c - np.cos(x64)).max() < 2.6e-7      # + one f32 rounding of the result
    hostlib.probe_soft(x.ctypes.data_as(fp), s.ctypes.data_as(fp), c.ctypes.data_as(fp), C.c_long(n))
    assert np.abs(s - np.sin(x64)).max() < 1.5e-7 and np.abs(c - np.cos(x64)).max() < 1.5e-7


@pytest.mark.gpu
def test_device_sincos_against_float64():
    import torch
    from brief_pytorch_amd import _lib
    x = torch.from_numpy(_inputs()).cuda()
    s, c = torch.empty_like(x), torch.empty_like(x)
    _lib.check(_lib.lib().brief_sincos_probe(_lib.ptr(x), _lib.ptr(s), _lib.ptr(c), x.numel(), _lib.stream_ptr()))
    x64 = x.double()
    es, ec = float((s.double() - torch.sin(x64)).abs().max()), float((c.double() - torch.cos(x64)).abs().max())
    print("device sin/cos max abs error vs float64: %.3g / %.3g" % (es, ec))
    assert es < 3.5e-7 and ec < 3.5e-7
    assert _lib.lib().brief_cu_count() >= 1
