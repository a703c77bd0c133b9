"""ctypes binding of libbrief_hip.so (include/brief_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails the
product raises.  PyTorch is used only for device memory and streams.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BRIEF_LIB", os.path.join(_HERE, "libbrief_hip.so"))   # BRIEF_LIB: A/B diagnostics only
SRC = os.path.join(_HERE, "csrc", "brief_hip.hip")
_CSRC = os.path.join(_HERE, "csrc")
# every file of the translation unit (brief_hip.hip includes the *.inc / *.h beside it) and the public header
# (an installation may ship only the prebuilt library, or point BRIEF_LIB at an external build: no csrc/ then, and nothing to rebuild)
_DEPS = sorted(os.path.join(_CSRC, f) for f in (os.listdir(_CSRC) if os.path.isdir(_CSRC) else []) if f.endswith((".hip", ".inc", ".h"))) \
    + [os.path.join(os.path.dirname(_HERE), "include", "brief_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


class BriefError(RuntimeError):
    pass


class SirenDesc(C.Structure):
    _fields_ = [("cin", C.c_int32), ("cout", C.c_int32), ("layers", C.c_int32), ("features", C.c_int32),
                ("w0_first", C.c_float), ("w0_hidden", C.c_float), ("output_act", C.c_int32), ("precision", C.c_int32)]


class GridDesc(C.Structure):
    _fields_ = [("ndim", C.c_int32), ("reserved", C.c_int32), ("dims", C.c_int64 * 3), ("lo", C.c_float), ("hi", C.c_float)]


class BatchDesc(C.Structure):
    _fields_ = [("coords", C.c_void_p), ("targets", C.c_void_p), ("weights", C.c_void_p), ("idx", C.c_void_p),
                ("offset", C.c_int64), ("n", C.c_int64), ("rng_pop", C.c_int64), ("rng_seed", C.c_uint64), ("rng_step", C.c_uint64)]


class FitJob(C.Structure):          # brief_fit_job
    _fields_ = [("desc", SirenDesc), ("grid", GridDesc), ("batch", BatchDesc),
                ("params", C.c_void_p), ("packed", C.c_void_p), ("state1", C.c_void_p), ("state2", C.c_void_p),
                ("grads", C.c_void_p), ("loss_out", C.c_void_p), ("loss_log", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
                ("loss_kind", C.c_int32), ("optim_kind", C.c_int32), ("thr", C.c_float), ("beta", C.c_float),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("milestones", C.POINTER(C.c_int64)), ("n_milestones", C.c_int32), ("reserved", C.c_int32),
                ("gamma", C.c_double), ("t0", C.c_int64),
                ("lr_table", C.POINTER(C.c_double)), ("beta1_table", C.POINTER(C.c_double)), ("idx_stride", C.c_int64)]


LOSS_KIND = {"datal2": 0, "datasmoothl1": 1, "external": 2}
OPT_KIND = {"Adamax": 0, "Adam": 1, "SGD": 2}
OUT_F32, OUT_U8, OUT_U16 = 0, 1, 2
PRECISION = {"fp32": 0, "f32": 0, "bf16": 1, "bf16x3": 2}

EXPORTS = ["brief_version", "brief_last_error", "brief_param_count", "brief_packed_count",
           "brief_train_workspace_bytes", "brief_siren_repack", "brief_siren_forward", "brief_forward_workspace_bytes", "brief_siren_forward_ws", "brief_siren_train_step", "brief_siren_fit_step",
           "brief_siren_fit", "brief_multi_fit",
           "brief_optim_step", "brief_sample_indices", "brief_sse_u16", "brief_profile_enable", "brief_profile_fused", "brief_deblock_edge", "brief_ssim_u16", "brief_ssim_partial_count",
           "brief_sincos_probe", "brief_cu_count"]


def needs_build():
    if LIB_PATH != os.path.join(_HERE, "libbrief_hip.so"):
        return False          # BRIEF_LIB names somebody else's build
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in _DEPS)


DIAG_LIB_PATH = os.path.join(_HERE, "libbrief_hip_diag.so")


def build(force=False, verbose=False, diagnostics=False, defines=(), out=None):
    """hipcc --offload-arch=gfx950 -> brief_pytorch_amd/libbrief_hip.so (in-tree).

    diagnostics=True builds libbrief_hip_diag.so with -DBRIEF_DIAGNOSTICS instead: the only build that reads the
    BRIEF_* environment knobs of tools/ (select it with BRIEF_LIB=...); the product library never calls getenv."""
    target = out or (DIAG_LIB_PATH if diagnostics else os.path.join(_HERE, "libbrief_hip.so"))
    if not force and not diagnostics and not defines and out is None and not needs_build():
        return LIB_PATH
    # -ffp-contract=off: every fused multiply-add in the kernels is an explicit fmaf/MFMA, so the
    # optimizer and de-normalise epilogues keep the separate roundings of the reference arithmetic
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", SRC, "-o", target]
    if diagnostics:
        cmd.append("-DBRIEF_DIAGNOSTICS")
    cmd += ["-D" + d for d in defines]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return target


_LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch first: libbrief_hip.so must bind to the HIP runtime torch already loaded, otherwise two
    # runtimes live in one process and ours sees no device
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise BriefError("libbrief_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                         "there is no CPU fallback for the fused SIREN path")
    L = C.CDLL(LIB_PATH)
    dp, gp, bp, vp = C.POINTER(SirenDesc), C.POINTER(GridDesc), C.POINTER(BatchDesc), C.c_void_p
    L.brief_version.restype = C.c_int
    if L.brief_version() != 130:
        raise BriefError("libbrief_hip.so is version %d, this binding needs 130 (brief_siren_forward_ws, widths to 4096): rebuild with __graft_entry__.build()" % L.brief_version())
    L.brief_last_error.restype = C.c_char_p
    L.brief_param_count.restype = C.c_int64
    L.brief_param_count.argtypes = [dp]
    L.brief_packed_count.restype = C.c_int64
    L.brief_packed_count.argtypes = [dp]
    L.brief_train_workspace_bytes.restype = C.c_int64
    L.brief_train_workspace_bytes.argtypes = [dp, C.c_int64]
    L.brief_siren_repack.argtypes = [dp, vp, vp, vp]
    L.brief_siren_forward.argtypes = [dp, vp, gp, bp, vp, C.c_int, C.c_float, C.c_float, C.c_double, C.c_double, vp]
    L.brief_forward_workspace_bytes.restype = C.c_int64
    L.brief_forward_workspace_bytes.argtypes = [dp, C.c_int64]
    L.brief_siren_forward_ws.argtypes = [dp, vp, gp, bp, vp, C.c_int, C.c_float, C.c_float, C.c_double, C.c_double, vp, C.c_int64, vp]
    L.brief_siren_train_step.argtypes = [dp, vp, gp, bp, C.c_int, C.c_float, C.c_float, vp, vp, vp, vp, C.c_int64, vp]
    L.brief_siren_fit_step.argtypes = [dp, vp, vp, gp, bp, C.c_int, C.c_float, C.c_float, C.c_int, vp, vp,
                                       C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64, vp, vp, vp, C.c_int64, vp]
    L.brief_siren_fit.argtypes = [C.POINTER(FitJob), C.c_int64, vp]
    L.brief_multi_fit.argtypes = [C.POINTER(FitJob), C.c_int32, C.c_int64, vp]
    L.brief_optim_step.argtypes = [C.c_int, vp, vp, vp, vp, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64, vp]
    L.brief_sample_indices.argtypes = [vp, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, vp]
    L.brief_sse_u16.argtypes = [vp, vp, C.c_int64, vp, vp]
    L.brief_deblock_edge.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64] + [C.c_int] * 6 + [C.c_double] * 3 + [C.c_int, vp]
    L.brief_ssim_partial_count.restype = C.c_int64
    L.brief_ssim_partial_count.argtypes = [C.c_int64] * 3
    L.brief_ssim_u16.argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_int64, vp, C.c_double, vp, C.c_int64, vp]
    L.brief_sincos_probe.argtypes = [vp, vp, vp, C.c_int64, vp]
    L.brief_cu_count.restype = C.c_int
    L.brief_profile_enable.argtypes = [C.c_int]
    L.brief_profile_fused.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    _LIB = L
    return L


def check(rc):
    if rc != 0:
        raise BriefError("libbrief_hip: status %d: %s" % (rc, lib().brief_last_error().decode()))


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())
