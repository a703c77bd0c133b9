#!/usr/bin/env python3
"""bench.py — encode throughput of BRIEF's SIREN fit loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json metric: "voxels/sec encode (512^3 vol, 4x256 SIREN)"):
  N=1  SingleTask: synthetic 512^3 uint16 volume, SIREN layers=5 features=256 ("4x256") w0=20,
       randompoint sampler sample_size=100000 (main.py:324-334: volumes > 80^3 use randompoint),
       datal2 loss, Adamax lr 1e-3 + MultiStepLR — the shipped default.yaml settings.
  N>1  DivideTask through the product path: NFGR.compress_divide on a (N*512, 512, 512) volume cut into N blocks of
       512^3 (divide_type total_N_1_1), one block and one 4x256 net per rank (independent units, main.py:547-575) —
       weak scaling, no collective on the data path; RCCL only for the barriers / max-time and the one
       [SSE, SSIM-sum, slices, voxels] all-reduce of the z-sharded evaluation.
A step = one pass of the hot path over one batch: sample 100000 voxels -> fused forward/loss/
backward -> optimizer update.  value = sampled voxels fitted per second over all ranks, with
the volume resident in HBM when the timed region starts.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a child `python -m
torch.distributed.run`, decided before anything touches the GPU) and exits with the child's code; under torchrun the
world size must equal --gpus.  The N = 1 line also carries, all timed in this run: `configs` (BASELINE config 1 on
k_small, config 3 on the bf16 kernels), a three-point `psnr_at_bitrate_sweep` on the 512^3 volume, and `encode` /
`decode` as wall clocks around a real NFGR.compress / NFGR.decompress of the host-resident volume.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from brief_pytorch_amd import _lib  # noqa: E402
from brief_pytorch_amd.fit import Fitter  # noqa: E402
from brief_pytorch_amd.networks import SIREN  # noqa: E402
from brief_pytorch_amd.synthetic import make_volume_torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz x 256 FLOP/clk
PEAK_BF16_MFMA_TFLOPS = 2500.0   # same guide: dense bf16 (v_mfma_f32_32x32x16_bf16); only used by --precision bf16
# --precision bf16x3 (BRIEF_PREC_BF16X3): every product is three bf16 MFMAs (hi*hi + hi*lo + lo*hi), so the algorithmic flops
# (counted once, as for f32) are priced against a third of the bf16 peak
PEAKS = {"fp32": PEAK_F32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "bf16x3": PEAK_BF16_MFMA_TFLOPS / 3.0}
DTYPES = {"fp32": "f32", "bf16": "bf16", "bf16x3": "bf16x3"}


def train_on_lean(F):
    """csrc/brief_layout.h: brief_use_lean(train): 3 tiles and every tile count from 5 to 32 except the 8-tile headline"""
    nt = (F + 31) // 32
    return ((nt >= 5 and nt != 8) or nt == 3) and nt <= 32


def fused_kernel_name(precision, F):
    if precision == "fp32" and F > 1024:
        nt = (F + 31) // 32
        m, P = min(((mm, (nt + 4 * mm - 1) // (4 * mm)) for mm in range(8, 2, -1)), key=lambda t: 4 * t[0] * t[1] * (1.0 + 0.015 * t[1]))      # csrc/brief_wide.inc: wide_mtw
        return "k_wide<%d,true> (%d feature tiles in %d passes, K-slabs staged from image-ordered planes)" % (m, nt, P)
    if precision == "fp32" and train_on_lean(F):
        nt = (F + 31) // 32
        return "k_lean<1,%d,0,true,%d> (%d feature tiles, run-time width)" % ((nt + 3) // 4, nt % 4, nt)
    if precision == "fp32":
        return "k_fused<%d,true>" % ((F + 31) // 32)
    if precision == "bf16x3":
        return "k_fused_x3<true> (split precision: fp16 / bf16 halves, 3 MFMAs per product, 64-sample tiles)"
    return "k16<%d,true,1>" % (F // 32)
LAYERS, FEATURES, W0, SAMPLE = 5, 256, 20.0, 100000      # BASELINE config 2 (the metric's shape); --config c3 = 8x512
BLOCK = (512, 512, 512)


def flops_per_sample(L, F, cin=3, cout=1):
    M = cin * F + (L - 2) * F * F + F * cout
    train = 2 * (3 * M - cin * F)
    fused = 2 * M + 2 * ((L - 2) * F * F + F * cout) + 2 * (cin * F + F * cout)     # fwd + dgrad + skinny wgrads
    return train, fused, 2 * M


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def _thread_counts():
    """the sweep's thread counts: oracle.thread_candidates() — the same affinity / cgroup-quota / pool-share rule the tests' oracle uses"""
    from oracle import oracle as O
    return O.thread_candidates()


def _oracle_rate(O, d, p, x, y, threads, max_steps, budget, warm=0):
    """voxels/s of the oracle's train step (fwd + loss + bwd + Adamax) on this batch with `threads` OpenMP threads (after `warm`
    untimed samples' worth of the same step: page faults of the activation buffers, thread start)"""
    O.lib().oracle_set_num_threads(int(threads))
    pp, s1, s2 = p.copy(), np.zeros_like(p), np.zeros_like(p)
    if warm:
        O.loss_grad(d, pp, x[:warm], y[:warm])
    t0, steps = time.perf_counter(), 0
    while True:
        _, g, _, _ = O.loss_grad(d, pp, x, y)
        O.optim_step("Adamax", pp, g, s1, s2, 1e-3, steps + 1)
        steps += 1
        el = time.perf_counter() - t0
        if el > budget or steps >= max_steps:
            return x.shape[0] * steps / el, steps


def cpu_baseline():
    """The CPU beside the GPU number, on this host, same step (4x256 SIREN, fwd + loss + bwd + Adamax): the
    oracle (a C port of the reference algorithm, OpenMP over samples) and a CPU-PyTorch autograd restatement of the reference's
    module.  EVERY figure — the thread sweep, the value, the one-thread rate — is timed on the REAL step, 100 000 samples, after a
    warm-up: round 4 swept on 20 000-sample steps, whose 100 MB of activations sit in the host's L3 while the real step's 512 MB do
    not — its sweep read 2.6x the rate its own `value` did.  `value` is the best sweep entry of the faster implementation."""
    from oracle import oracle as O
    d = O.make_desc(3, 1, LAYERS, FEATURES, W0)
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    p = SIREN(features=FEATURES, layers=LAYERS, w0=W0).params.numpy().copy()
    n = SAMPLE
    x = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    y = rng.uniform(0, 100, size=(n, 1)).astype(np.float32)
    ncpu = os.cpu_count() or 1
    threads0 = O.lib().oracle_num_threads()
    sweep = {}
    for t in _thread_counts():
        sweep[t] = _oracle_rate(O, d, p, x, y, t, 2, 3.0, warm=20000)[0]
    best_t = max(sweep, key=sweep.get)
    rate = sweep[best_t]
    one = _oracle_rate(O, d, p, x, y, 1, 1, 0.0, warm=2000)[0]
    O.lib().oracle_set_num_threads(int(threads0))
    sample = ("thread sweep %s, each entry 1-2 steps of %d samples after a 20000-sample warm-up; value = its best entry; one_thread: one %d-sample step; "
              "oracle/siren_oracle.c (OpenMP over samples)" % (sorted(sweep), n, n))
    oracle_out = {"value": rate, "unit": "voxels/s", "threads": best_t, "one_thread": one,
                  "thread_sweep": {str(k): v for k, v in sorted(sweep.items())}, "sample": sample}
    out = {"value": rate, "unit": "voxels/s", "cores": best_t, "kind": "port", "host_cpus": ncpu, "cpu_share": O.cpu_share(), "cpu_model": cpu_model(),
           "sample": sample, "one_thread": one, "oracle": oracle_out}
    try:
        tc = torch_cpu_baseline(x, y, p)
        out["torch_cpu"] = tc
        if tc["value"] > out["value"]:
            out.update({"value": tc["value"], "cores": tc["threads"], "sample": tc["sample"], "one_thread": tc["one_thread"]})
    except Exception as e:      # informational only: the oracle line above is the baseline
        out["torch_cpu"] = {"error": repr(e)[:120]}
    return out


def torch_cpu_baseline(x, y, p):
    """The same step as the reference runs it (nn.Linear + sin, F.mse_loss, autograd, torch.optim.Adamax:
    utils/Networks.py:246-271, main.py:385-400), restated here and timed with CPU PyTorch on this host: a thread sweep and one
    thread, every entry on the full 100 000-sample step after a warm-up step."""
    import torch.nn as nn

    class Sine(nn.Module):
        def __init__(self, w0):
            super().__init__()
            self.w0 = w0

        def forward(self, v):
            return torch.sin(self.w0 * v)

    def make():
        layers = [nn.Linear(3, FEATURES), Sine(W0)]
        for _ in range(LAYERS - 2):
            layers += [nn.Linear(FEATURES, FEATURES), Sine(30.0)]
        layers += [nn.Linear(FEATURES, 1)]
        net = nn.Sequential(*layers)
        with torch.no_grad():       # same parameters as the oracle run (canonical order: W, b per layer)
            off = 0
            for m in net:
                if isinstance(m, nn.Linear):
                    nw = m.weight.numel()
                    m.weight.copy_(torch.from_numpy(p[off:off + nw].reshape(m.weight.shape))); off += nw
                    m.bias.copy_(torch.from_numpy(p[off:off + m.bias.numel()])); off += m.bias.numel()
        return net, torch.optim.Adamax(net.parameters(), lr=1e-3)

    def rate(xt, yt, threads, max_steps, budget):
        torch.set_num_threads(int(threads))
        net, opt = make()
        steps, t0 = 0, None
        while True:
            opt.zero_grad()
            loss = torch.nn.functional.mse_loss(net(xt), yt)
            loss.backward()
            opt.step()
            if t0 is None:
                t0 = time.perf_counter()        # first step is the warm-up
                continue
            steps += 1
            el = time.perf_counter() - t0
            if el > budget or steps >= max_steps:
                return xt.shape[0] * steps / el, steps

    threads0 = torch.get_num_threads()
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    sweep = {t: rate(xt, yt, t, 2, 3.0)[0] for t in _thread_counts()}
    best_t = max(sweep, key=sweep.get)
    one = rate(xt, yt, 1, 1, 0.0)[0]
    torch.set_num_threads(threads0)
    return {"value": sweep[best_t], "unit": "voxels/s", "threads": best_t, "one_thread": one,
            "thread_sweep": {str(k): v for k, v in sorted(sweep.items())},
            "sample": "thread sweep %s, each entry 1-2 steps of %d samples after one warm-up step; value = its best entry; one_thread: one such step after a "
                      "warm-up; torch %s CPU autograd" % (sorted(sweep), x.shape[0], torch.__version__)}


def divide_bench(args, dist, rank, world, dev, red_dev):
    """N > 1: the timed steps run INSIDE NFGR.compress_divide (brief_pytorch_amd/framework.py): rank 0 partitions and
    broadcasts, every rank fits the block it owns from a memory map of the shared volume file, the evaluation is sharded
    by z.  The fit pauses at two step marks (after pre-roll + warm-up, and K steps later) where this function brackets the
    timed region with barrier + synchronize."""
    import shutil
    import tempfile
    from brief_pytorch_amd import config
    from brief_pytorch_amd.dist_utils import broadcast_object
    from brief_pytorch_amd.framework import NFGR, MyLogger
    from brief_pytorch_amd.tool import create_stack, write_slab
    L = _lib.lib()
    work = broadcast_object(tempfile.mkdtemp(prefix="brief_bench_") if rank == 0 else None)
    path = os.path.join(work, "volume.npy")
    shape = (world * BLOCK[0], BLOCK[1], BLOCK[2], 1)
    if rank == 0:
        create_stack(path, shape, np.uint16)
    dist.barrier()
    blk = make_volume_torch(BLOCK, seed=42 + rank, device=dev)
    write_slab(path, rank * BLOCK[0], blk.cpu().numpy())           # every rank writes its own block of the shared file
    del blk
    torch.cuda.empty_cache()
    dist.barrier()
    opt = config.load(os.path.join(ROOT, "opt", "DivideTask", "default.yaml"))
    cf = opt.CompressFramework
    pre = args.preroll
    total_steps = pre + args.warmup + args.steps
    pcount = SIREN.calc_param_count(3, 1, FEATURES, LAYERS)
    cf.Compress.divide.divide_type = "total_%d_1_1" % world
    cf.Compress.divide.param_alloc = "by_size"
    cf.Compress.param.filesize_ratio, cf.Compress.param.given_size = 0, 4.0 * pcount * world
    cf.Compress.max_steps, cf.Compress.checkpoints = total_steps, "none"
    cf.Compress.sampler.name, cf.Compress.sampler.sample_size = "randompoint", SAMPLE
    cf.Compress.loss_log_freq = 10 ** 9
    cf.Compress.decompress = not args.no_psnr
    cf.Module.phi.layers, cf.Module.phi.w0 = LAYERS, W0
    cf.Compress.precision = args.precision
    cf["_seed"] = 42
    cf.Decompress.keep_decompressed, cf.Decompress.mip, cf.Decompress.ssim = False, False, True
    Log = MyLogger(outputs_dir=work, project_name="run", time=False, logdir=os.path.join(work, "run"))
    torch.manual_seed(42)
    fw = NFGR(cf, Log=Log)
    marks = [pre + args.warmup, total_steps]
    stamp = {}

    _lib.check(L.brief_profile_enable(1))         # creates the library's timing events (tens of ms on the host): outside the timed window
    _lib.check(L.brief_profile_enable(0))

    def on_mark(k):
        dist.barrier()
        torch.cuda.synchronize()
        if k == marks[0]:
            _lib.check(L.brief_profile_enable(1))      # (cheap now: resets the slot counter)
        stamp[k] = time.perf_counter()
    res = fw.compress_divide(path, opt, marks=marks, on_mark=on_mark)
    elapsed = stamp[marks[1]] - stamp[marks[0]]
    tot_ms, launches = C.c_double(0), C.c_int64(0)
    _lib.check(L.brief_profile_fused(C.byref(tot_ms), C.byref(launches)))
    _lib.check(L.brief_profile_enable(0))
    tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    perf = res.get(total_steps, {}) if rank == 0 else {}
    dist.barrier()
    if rank == 0:
        shutil.rmtree(work, ignore_errors=True)
    return elapsed, tot_ms.value / max(launches.value, 1), perf, pcount


def launch_ranks(args):
    """--gpus N without a launcher: start N ranks as a CHILD process (never an exec: the box forbids replacing a process
    that touched the GPU, and this one has not touched it yet either) and hand its exit code back"""
    import socket
    import subprocess
    shared = os.environ.get("BRIEF_DIST_BACKEND", "nccl") == "gloo" and os.environ.get("BRIEF_SHARE_GPU") == "1"      # rehearsal: ranks share devices
    if args.gpus > torch.cuda.device_count() and not shared:           # (the ranks are a CHILD process started below, never an exec of this one)
        sys.stderr.write("bench.py: --gpus %d but only %d device(s) are visible\n" % (args.gpus, torch.cuda.device_count()))
        sys.exit(2)
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def timed_config(name, L, F, dims, sampler, n, precision, steps, tgt=None, seed=7, decode=False):
    """one BASELINE configuration on this GPU: `steps` optimizer steps in one brief_siren_fit call, wall clock bracketed by
    synchronize, the dominant kernel timed by HIP events inside the library (as for the headline)."""
    L_ = _lib.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    if tgt is None:
        vol = make_volume_torch(dims, seed=seed, device=dev)
        t = vol.view(-1, 1).to(torch.float32)
        vmin, vmax = float(t.min().item()), float(t.max().item())
        tgt = (t - np.float32(vmin)) / np.float32(vmax - vmin)
        tgt *= np.float32(100.0)
    torch.manual_seed(seed)
    net = SIREN(coords_channel=3, data_channel=1, features=F, layers=L, w0=W0, precision=precision).to(dev)
    fit = Fitter(net, tgt, dims, sampler=sampler, sample_size=n, optimizer="Adamax", lr=1e-3,
                 scheduler={"name": "MultiStepLR", "milestones": [50000, 60000, 70000], "gamma": 0.2}, seed=seed)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:               # clock ramp + caches
        fit.run(20)
        torch.cuda.synchronize()
    _lib.check(L_.brief_profile_enable(1))
    t1 = time.perf_counter()
    fit.run(steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t1
    tot_ms, launches = C.c_double(0), C.c_int64(0)
    _lib.check(L_.brief_profile_fused(C.byref(tot_ms), C.byref(launches)))
    _lib.check(L_.brief_profile_enable(0))
    train_f, fused_f, _ = flops_per_sample(L, F)
    peak = PEAKS[precision]
    nb = fit.n
    kms = tot_ms.value / max(launches.value, 1)
    small = F <= 64 and precision == "fp32"              # k_small is the whole train step but the reduction
    kflop = (train_f if small else fused_f) * nb
    dec = {}
    if decode:
        # decode_grid of the whole block with this net (output left in HBM, de-normalise + uint16 cast fused), as for the headline
        out = net.decode_grid(dims, out_kind="u16", scale=(0.0, 100.0), vrange=(0.0, 65535.0))
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        net.decode_grid(dims, out_kind="u16", scale=(0.0, 100.0), vrange=(0.0, 65535.0), out=out)
        torch.cuda.synchronize()
        t_dec = time.perf_counter() - t2
        nvox = float(np.prod(dims))
        dec = {"decode_kernel": {"seconds": t_dec, "voxels_per_s": nvox / t_dec, "tflops": nvox * flops_per_sample(L, F)[2] / t_dec / 1e12}}
        del out
    return {**dec, "workload": name, "layers": L, "features": F, "volume": list(dims), "samples_per_step": nb, "dtype": DTYPES[precision],
            "steps": steps, "ms_per_step": el * 1e3 / steps, "voxels_per_s": nb * steps / el,
            "step_tflops": train_f * nb / (el / steps) / 1e12, "step_frac": train_f * nb / (el / steps) / 1e12 / peak,
            "kernel": "k_small" if small else {"fp32": "k_wide" if F > 1024 else ("k_lean" if train_on_lean(F) else "k_fused"), "bf16x3": "k_fused_x3<true>", "bf16": "k16 (body + tail launches)"}[precision],
            "kernel_ms": kms, "kernel_tflops": kflop / (kms * 1e-3) / 1e12, "kernel_frac": kflop / (kms * 1e-3) / 1e12 / peak, "peak_tflops": peak}


def device_identity(dev):
    """what proves N ranks ran on N devices: the device's UUID (hipDeviceProp.uuid through torch), name, index and CU count"""
    pr = torch.cuda.get_device_properties(dev)
    uuid = getattr(pr, "uuid", None)
    pci = [getattr(pr, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
    return {"index": int(dev.index if dev.index is not None else torch.cuda.current_device()), "uuid": str(uuid) if uuid is not None else None,
            "pci": ":".join("%x" % v for v in pci) if None not in pci else None,
            "name": pr.name, "compute_units": int(pr.multi_processor_count), "visible_devices": torch.cuda.device_count(),
            "env": {k: os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES") if os.environ.get(k) is not None}}


def gather_ranks(dist, rank, world, dev, backend):
    """per-rank identity, gathered on every rank: rank, local rank, the process group's world size and backend, the device"""
    me = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "world_size": world, "backend": backend, "pid": os.getpid(), "device": device_identity(dev)}
    if dist is None or world == 1:
        return [me]
    allr = [None] * world
    dist.all_gather_object(allr, me)
    return allr


def summarize(out):
    """flat scalars the driver's parser keeps: one entry per secondary number of this line"""
    sm = {}
    for k, c in (out.get("configs") or {}).items():
        sm[k + ".ms_per_step"] = round(c["ms_per_step"], 5)
        sm[k + ".step_frac"] = round(c["step_frac"], 4)
        sm[k + ".kernel_frac"] = round(c["kernel_frac"], 4)
    for k in ("encode", "decode"):
        if k in out:
            sm[k + ".wall_seconds"] = round(out[k]["wall_seconds"], 4)
            sm[k + ".voxels_per_s"] = round(out[k]["voxels_per_s"], 1)
    if "decode_kernel" in out:
        sm["decode_kernel.seconds"] = round(out["decode_kernel"]["seconds"], 5)
        sm["decode_kernel.tflops"] = round(out["decode_kernel"]["tflops"], 2)
    for pt in out.get("psnr_at_bitrate_sweep") or []:
        sm["psnr_clean_db@%dsteps.F%d" % (pt["steps"], pt["features"])] = round(pt["psnr_clean_db"], 3)
    if "psnr_at_bitrate_20000" in out:
        q = out["psnr_at_bitrate_20000"]
        sm["psnr_db@20000"], sm["psnr_clean_db@20000"], sm["ssim@20000"] = round(q["psnr_db"], 3), round(q["psnr_clean_db"], 3), round(q.get("ssim", float("nan")), 5)
    return sm


def psnr_u16(a, b):
    """cal_psnr (utils/misc.py:451-456) of two uint16 volumes on the device, from the integer SSE"""
    sse = torch.zeros(1, dtype=torch.float64, device=a.device)
    _lib.check(_lib.lib().brief_sse_u16(_lib.ptr(a), _lib.ptr(b), a.numel(), _lib.ptr(sse), _lib.stream_ptr()))
    return float(-10.0 * np.log10(max(sse.item(), 1e-30) / a.numel() / 65535.0 ** 2))


def quality(vol, clean, dec, ssim=False):
    """PSNR against the source (what the reference reports: the source carries N(0, 200) noise, which caps this figure at
    20 log10(65535 / 200) = 50.3 dB whatever the codec does) AND against the noise-free field the source was generated from
    (what tells a good fit from a bad one); SSIM (utils/misc.py:458-475) against the source on request."""
    out = {"psnr_db": psnr_u16(vol, dec), "psnr_clean_db": psnr_u16(clean, dec)}
    if ssim:
        from brief_pytorch_amd.metrics import gpu_ssim_u16
        tot, slices = gpu_ssim_u16(vol, dec)
        out["ssim"] = float(tot) / float(slices)
    return out


def rate_point(F, tgt, vol, clean, vmin, vmax, steps, seed=42):
    """one point of the PSNR-vs-bitrate curve on the 512^3 block: a (LAYERS, F) net fitted for `steps` steps, decoded, PSNR from
    the GPU SSE"""
    dev = tgt.device
    torch.manual_seed(seed)
    net = SIREN(coords_channel=3, data_channel=1, features=F, layers=LAYERS, w0=W0).to(dev)
    fit = Fitter(net, tgt, BLOCK, sampler="randompoint", sample_size=SAMPLE, optimizer="Adamax", lr=1e-3,
                 scheduler={"name": "MultiStepLR", "milestones": [50000, 60000, 70000], "gamma": 0.2}, seed=seed)
    t0 = time.perf_counter()
    fit.run(steps)
    dec = net.decode_grid(BLOCK, out_kind="u16", scale=(0.0, 100.0), vrange=(vmin, vmax))
    q = quality(vol, clean, dec)
    return {"features": F, "params": net.param_count, "bits_per_voxel": 32.0 * net.param_count / float(np.prod(BLOCK)), "steps": steps,
            **q, "seconds": time.perf_counter() - t0}


def encode_decode_wall(vol_host, steps, precision="fp32"):
    """SURVEY 8d's encode figure, measured: wall clock around NFGR.compress of the HOST-resident volume (preprocess, the
    reference's `_preprocessed` dump, loss-weight map, normalise, H2D, net init, `steps` optimizer steps, weight files +
    sideinfos.yaml) and around NFGR.decompress of the stored artefact (load, decode kernel, D2H, postprocess)."""
    import shutil
    import tempfile
    from brief_pytorch_amd import config
    from brief_pytorch_amd.framework import NFGR, MyLogger, _wrap
    work = tempfile.mkdtemp(prefix="brief_e2e_")
    try:
        opt = config.load(os.path.join(ROOT, "opt", "SingleTask", "default.yaml"))
        cf = opt.CompressFramework
        pcount = SIREN.calc_param_count(3, 1, FEATURES, LAYERS)
        cf.Module.phi.layers, cf.Module.phi.w0 = LAYERS, W0
        cf.Compress.param.filesize_ratio, cf.Compress.param.given_size = 0, 4.0 * pcount
        cf.Compress.max_steps, cf.Compress.checkpoints, cf.Compress.loss_log_freq = steps, "none", 10 ** 9
        cf.Compress.decompress = False
        cf.Compress.precision = precision
        cf["_seed"] = 42
        Log = MyLogger(outputs_dir=work, project_name="e2e", time=False)
        torch.manual_seed(42)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fw = NFGR(cf, Log=Log)
        fw.compress(os.path.join(work, "volume.npy"), data=vol_host)
        torch.cuda.synchronize()
        t_enc = time.perf_counter() - t0
        cdir = os.path.join(Log.logdir, "steps%d" % steps, "compressed")
        t1 = time.perf_counter()
        dec = NFGR.decompress(opt, os.path.join(cdir, "module"), os.path.join(cdir, "sideinfos.yaml"))
        t_dec = time.perf_counter() - t1
        d = dec.astype(np.float32) - vol_host.astype(np.float32)
        psnr = float(-10.0 * np.log10(float((d.astype(np.float64) ** 2).mean()) / 65535.0 ** 2))
        files = sum(os.path.getsize(os.path.join(cdir, "module", f)) for f in os.listdir(os.path.join(cdir, "module")))
        return {"steps": steps, "wall_seconds": t_enc, "voxels_per_s": vol_host.size / t_enc, "fit_seconds": fw.fit_seconds,
                "artefact_bytes": files + os.path.getsize(os.path.join(cdir, "sideinfos.yaml")),
                "note": "wall clock of NFGR.compress on the host-resident 512^3 uint16 volume: preprocess + _preprocessed dump + normalise + H2D + "
                        "net init + %d steps (fit_seconds) + weight files" % steps}, \
               {"wall_seconds": t_dec, "voxels_per_s": vol_host.size / t_dec, "psnr_db": psnr,
                "note": "wall clock of NFGR.decompress of the stored artefact: load weights, decode kernel (de-normalise + cast fused), D2H, postprocess"}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--preroll", type=int, default=200, help="untimed steps before the counted warm-up (clock ramp)")
    ap.add_argument("--preroll-seconds", type=float, default=0.5, help="... and at least this long")
    ap.add_argument("--encode-steps", type=int, default=2000, help="total optimizer steps of the end-to-end encode figure / psnr_at_bitrate (0: skip)")
    ap.add_argument("--long-steps", type=int, default=20000, help="the reference's own schedule length (default.yaml max_steps): PSNR / SSIM point after this many steps (0: skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-psnr", action="store_true")
    ap.add_argument("--precision", choices=["fp32", "bf16", "bf16x3"], default="fp32",
                    help="fp32 = the metric (exact f32 MFMA); bf16 = BASELINE config 3's arithmetic; bf16x3 = split-precision bf16 MFMAs that "
                         "meet the fp32 parity bands (extra lines, never the default)")
    ap.add_argument("--config", choices=["c2", "c3"], default="c2", help="c2: 4x256 SIREN (the metric); c3: 8x512 SIREN")
    ap.add_argument("--divide", action="store_true", help="run the DivideTask product path also at world size 1 (one block, process group initialised)")
    ap.add_argument("--block", type=int, default=512, help="edge of the cubic block (tests only; the metric is quoted on 512)")
    ap.add_argument("--no-extras", action="store_true", help="skip configs / psnr_at_bitrate_sweep / the NFGR wall clocks (headline line only)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)                              # does not return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus\n" % (args.gpus, world))
        sys.exit(2)
    global BLOCK
    BLOCK = (args.block,) * 3
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # "nccl" IS RCCL on ROCm.  BRIEF_DIST_BACKEND=gloo only exists to rehearse the N>1 code path on a
    # box with fewer GPUs than ranks (ranks then share devices and the reductions go through the host).
    backend = os.environ.get("BRIEF_DIST_BACKEND", "nccl")
    if world > 1 or args.divide:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
        if world == 1:                                   # --divide without a launcher: a one-rank group
            import socket
            sock = socket.socket()
            sock.bind(("127.0.0.1", 0))
            free_port = sock.getsockname()[1]
            sock.close()
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank % torch.cuda.device_count()))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    _lib.lib()   # fails loudly when the HIP extension is missing
    global LAYERS, FEATURES
    if args.config == "c3":
        LAYERS, FEATURES = 9, 512

    if world > 1 or args.divide:
        if backend == "nccl" and world > torch.cuda.device_count():
            sys.stderr.write("bench.py: %d ranks but %d visible device(s): one rank per GPU is the contract\n" % (world, torch.cuda.device_count()))
            sys.exit(2)
        ranks = gather_ranks(dist, rank, world, dev, backend)
        # what identifies a rank's device: (local index, UUID, PCI address) — reported, never a reason to stop (a runtime that leaves the UUID
        # empty must not fail a legitimate 8-GPU run; the index alone is distinct by construction: local_rank < device_count)
        uuids = [(r["device"]["index"], r["device"]["uuid"], r["device"]["pci"]) for r in ranks]
        elapsed, fused_ms, perf, pcount = divide_bench(args, dist, rank, world, dev, red_dev)
        if rank == 0:
            train_f, fused_f, _ = flops_per_sample(LAYERS, FEATURES)
            peak = PEAKS[args.precision]
            ms_step = elapsed * 1e3 / args.steps
            achieved = fused_f * SAMPLE / (fused_ms * 1e-3) / 1e12
            out = {
                "metric": "encode_voxels_per_sec", "value": SAMPLE * args.steps * world / elapsed, "unit": "voxels/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": DTYPES[args.precision], "data": "synthetic",
                "config": {"workload": "DivideTask (NFGR.compress_divide) on a synthetic uint16 volume of %d x %d^3 blocks (total_%d_1_1), one block and "
                                       "one SIREN %dx%d (layers=%d, features=%d, w0=20) per rank, randompoint sample_size=100000, datal2, Adamax lr=1e-3"
                                       % (world, BLOCK[0], world, LAYERS - 1, FEATURES, LAYERS, FEATURES),
                           "volume": [world * BLOCK[0], BLOCK[1], BLOCK[2]], "layers": LAYERS, "features": FEATURES, "sample_size": SAMPLE,
                           "params": pcount, "bits_per_voxel": 32.0 * pcount / float(np.prod(BLOCK))},
                "distinct_devices": len(set(uuids)), "backend": backend,
                "roofline": {"bound": "mfma", "kernel": fused_kernel_name(args.precision, FEATURES)
                             + " (forward+loss+dgrad), rank 0", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                             "traffic": None, "traffic_source": None, "kernel_ms": fused_ms, "flop_per_launch": fused_f * SAMPLE,
                             "step_tflops": train_f * SAMPLE / (ms_step * 1e-3) / 1e12, "step_frac": train_f * SAMPLE / (ms_step * 1e-3) / 1e12 / peak},
                "preroll_steps": args.preroll, "ranks": ranks,
                "psnr_at_bitrate": {"steps": args.preroll + args.warmup + args.steps, "bits_per_voxel": 32.0 * pcount / float(np.prod(BLOCK)),
                                    "psnr_db": perf.get("psnr"), "ssim": perf.get("ssim"),
                                    "note": "merged volume, z-sharded decode, [SSE, SSIM-sum, slices, voxels] all-reduced over the ranks"},
            }
            print(json.dumps(out), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return

    # ---- data: this rank's block, generated and normalised on the device (utils/io.py:65-80 op order)
    vol = make_volume_torch(BLOCK, seed=42 + rank, device=dev)
    tgt = vol.view(-1, 1).to(torch.float32)     # (torch has no uint16 min/max kernels)
    vmin, vmax = float(tgt.min().item()), float(tgt.max().item())
    tgt = (tgt - np.float32(vmin)) / np.float32(vmax - vmin)
    tgt *= np.float32(100.0)
    tgt += np.float32(0.0)
    torch.manual_seed(42)
    net = SIREN(coords_channel=3, data_channel=1, features=FEATURES, layers=LAYERS, w0=W0, precision=args.precision).to(dev)
    fit = Fitter(net, tgt, BLOCK, sampler="randompoint", sample_size=SAMPLE, optimizer="Adamax", lr=1e-3,
                 scheduler={"name": "MultiStepLR", "milestones": [50000, 60000, 70000], "gamma": 0.2}, seed=42 + rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # the library creates its timing events on the first enable (8192 hipEventCreate calls: tens of ms on the host): do that
    # NOW, not between the warm-up and the timed steps, where the idle device would drop its clocks and the short timed window
    # of the driver's flags (20 steps = 20 ms) would run inside the ramp back up (measured: 1.005-1.014 ms per step against 0.987
    # for a 200-step window and 0.981 in steady state)
    _lib.check(_lib.lib().brief_profile_enable(1))
    _lib.check(_lib.lib().brief_profile_enable(0))
    # untimed pre-roll: a fresh device needs a few hundred ms of load before its clocks and caches are steady; without
    # it a short timed window (the driver's --steps 20 --warmup 5 = 30 ms) measures the ramp, not the kernel
    t_pre = time.perf_counter()
    pre = 0
    while pre < args.preroll or time.perf_counter() - t_pre < args.preroll_seconds:
        fit.step()
        pre += 1
        if pre % 50 == 0:
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        fit.step()
    L = _lib.lib()
    barrier()
    _lib.check(L.brief_profile_enable(1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = fit.step()
    barrier()
    elapsed = time.perf_counter() - t0
    tot_ms, launches = C.c_double(0), C.c_int64(0)
    _lib.check(L.brief_profile_fused(C.byref(tot_ms), C.byref(launches)))
    _lib.check(L.brief_profile_enable(0))
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- end-to-end figures: continue the SAME fit to --encode-steps optimizer steps in one C-ABI call,
    #      decode the whole block with the forward kernel (de-normalise + cast fused), PSNR from the GPU SSE
    psnr, extra, q2k = None, {}, {}
    steps_done = pre + args.warmup + args.steps
    if not args.no_psnr:
        more = max(args.encode_steps - steps_done, 0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if more:
            fit.run(more)
        torch.cuda.synchronize()
        t_more = time.perf_counter() - t1
        steps_done += more
        t2 = time.perf_counter()
        dec = net.decode_grid(BLOCK, out_kind="u16", scale=(0.0, 100.0), vrange=(vmin, vmax))
        torch.cuda.synchronize()
        t_dec = time.perf_counter() - t2
        sse = torch.zeros(1, dtype=torch.float64, device=dev)
        _lib.check(L.brief_sse_u16(_lib.ptr(vol), _lib.ptr(dec), vol.numel(), _lib.ptr(sse), _lib.stream_ptr()))
        red = torch.tensor([sse.item(), float(vol.numel())], dtype=torch.float64, device=red_dev)
        if dist is not None:
            dist.all_reduce(red)          # the one collective of the DivideTask path: [SSE, n]
        psnr = float(-10.0 * np.log10(red[0].item() / red[1].item() / 65535.0 ** 2))
        clean = make_volume_torch(BLOCK, seed=42 + rank, device=dev, noise_sigma=0.0)      # the same field without its N(0, 200) noise
        q2k = quality(vol, clean, dec, ssim=True)
        nvox = float(np.prod(BLOCK))
        ms_more = t_more * 1e3 / more if more else elapsed * 1e3 / args.steps
        extra = {"decode_kernel": {"seconds": t_dec, "voxels_per_s": nvox / t_dec, "tflops": nvox * flops_per_sample(LAYERS, FEATURES)[2] / t_dec / 1e12,
                                   "note": "decode_grid of the whole block, output left in HBM: coordinates synthesised in-kernel, de-normalise + uint16 cast fused"},
                 "steady_state_fit": {"steps": steps_done, "ms_per_step": ms_more, "note": "continuation of the timed fit to --encode-steps in one brief_siren_fit call"}}
        if not args.no_extras and world == 1:
            # ---- the other BASELINE configurations, driver-timed in this run
            cfgs = {}
            cfgs["c1_64cube_2x64_full_batch"] = timed_config("SingleTask 64^3 synthetic volume, SIREN 2x64 (layers=3, features=64), one randomcube window = "
                                                             "the whole grid (262144 samples per step), Adamax", 3, 64, (64, 64, 64), "full", 0, "fp32", 400)
            cfgs["default_yaml_64cube_4x22"] = timed_config("SingleTask default.yaml on a 64^3 volume: the budget solves to SIREN 4x22 (layers=5, features=22), "
                                                            "full-volume batch", 5, 22, (64, 64, 64), "full", 0, "fp32", 400)
            # the shipped default.yaml over volume sizes: its byte budget (ratio 80) solves to 22 / 65 / 186 / 527 features on 64^3 / 128^3 / 256^3 / 512^3
            # uint16 volumes (SIREN.calc_features), i.e. k_small, k_lean with one, two and five slots per wave
            cfgs["default_yaml_128cube_4x65"] = timed_config("SingleTask default.yaml on a 128^3 uint16 volume: the budget solves to SIREN 4x65 (3 feature tiles), "
                                                             "randompoint sample_size=100000; k_lean<1,1,0,true,3> + k_wgrad<3>", 5, 65, (128, 128, 128), "randompoint", SAMPLE, "fp32", 200)
            cfgs["default_yaml_256cube_4x186"] = timed_config("SingleTask default.yaml on a 256^3 uint16 volume: the budget solves to SIREN 4x186 (6 feature tiles), "
                                                              "randompoint sample_size=100000; k_lean + k_wgrad<6>", 5, 186, (256, 256, 256), "randompoint", SAMPLE, "fp32", 100)
            cfgs["default_yaml_512cube_4x527"] = timed_config("SingleTask default.yaml (ratio 80) on the 512^3 uint16 volume: the budget solves to SIREN 4x527 "
                                                              "(layers=5, features=527 = 17 feature tiles), randompoint sample_size=100000; k_lean + k_wgrad<0,6>",
                                                              5, 527, BLOCK, "randompoint", SAMPLE, "fp32", 40, tgt=tgt)
            cfgs["default_yaml_1024cube_4x1494"] = timed_config("the net opt/SingleTask/default.yaml (ratio 80) solves to on a 1024^3 uint16 volume: SIREN 4x1494 (layers=5, "
                                                               "features=1494 = 47 feature tiles), randompoint sample_size=100000, sampled on this bench's 512^3 volume; "
                                                               "k_wide<6,true> + k_wgrad<0,8>", 5, 1494, BLOCK, "randompoint", SAMPLE, "fp32", 10, tgt=tgt)
            cfgs["c3_512cube_8x512_bf16"] = timed_config("SingleTask 512^3 synthetic volume, SIREN 8x512 (layers=9, features=512), bf16 MFMA with f32 master weights, "
                                                         "randompoint sample_size=100000", 9, 512, BLOCK, "randompoint", SAMPLE, "bf16", 60, tgt=tgt)
            cfgs["c2_512cube_4x256_bf16x3"] = timed_config("the headline workload (SingleTask 512^3, SIREN 4x256, randompoint sample_size=100000) under precision="
                                                           "bf16x3: hidden GEMMs as three bf16 MFMAs per product on hi/lo operand halves, f32 accumulate and stashes; "
                                                           "passes the fp32 parity bands (tests/test_gpu_bf16x3.py); priced against bf16 peak / 3; not the metric",
                                                           LAYERS, FEATURES, BLOCK, "randompoint", SAMPLE, "bf16x3", 200, tgt=tgt, decode=True)
            extra["configs"] = cfgs
            # ---- PSNR against bitrate on the 512^3 volume: three net sizes, --encode-steps steps each
            pts = [rate_point(F_, tgt, vol, clean, vmin, vmax, steps_done) for F_ in (64, 128)]
            pts.append({"features": FEATURES, "params": net.param_count, "bits_per_voxel": 32.0 * net.param_count / nvox, "steps": steps_done, **q2k})
            pts += [rate_point(F_, tgt, vol, clean, vmin, vmax, steps_done) for F_ in (384, 512, 640)]
            extra["psnr_at_bitrate_sweep"] = pts
            extra["psnr_at_bitrate_sweep_note"] = ("psnr_db: against the source volume, whose N(0, 200) noise caps it at 50.3 dB for any codec; psnr_clean_db: against "
                                                   "the noise-free field the source was generated from (make_volume_torch(noise_sigma=0)) — the figure that separates the widths")
            # ---- the reference's own schedule length (opt/SingleTask/default.yaml:40 max_steps 20000): the headline fit continued to
            #      --long-steps optimizer steps, decoded, PSNR / SSIM
            if args.long_steps > steps_done:
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                fit.run(args.long_steps - steps_done)
                torch.cuda.synchronize()
                t_long = time.perf_counter() - t3
                dec_l = net.decode_grid(BLOCK, out_kind="u16", scale=(0.0, 100.0), vrange=(vmin, vmax))
                extra["psnr_at_bitrate_20000"] = {"steps": args.long_steps, "bits_per_voxel": 32.0 * net.param_count / nvox, **quality(vol, clean, dec_l, ssim=True),
                                                  "fit_seconds_for_the_last_%d_steps" % (args.long_steps - steps_done): t_long,
                                                  "ms_per_step": t_long * 1e3 / (args.long_steps - steps_done),
                                                  "encode_voxels_per_s_at_this_length": nvox / (t_long * args.long_steps / (args.long_steps - steps_done))}
                del dec_l
            # ---- encode / decode as wall clocks of the product path on the host-resident volume
            vol_h = vol.cpu().numpy()
            enc, dec_w = encode_decode_wall(vol_h, args.encode_steps if args.encode_steps > 0 else 2000)
            extra["encode"], extra["decode"] = enc, dec_w
            # the same two wall clocks with Compress.precision: bf16x3 (same seed, same steps; never the metric)
            enc3, dec3 = encode_decode_wall(vol_h, args.encode_steps if args.encode_steps > 0 else 2000, "bf16x3")
            cfgs["c2_512cube_4x256_bf16x3"]["encode"], cfgs["c2_512cube_4x256_bf16x3"]["decode"] = enc3, dec3

    q2k_out = {k: v for k, v in q2k.items() if k != "psnr_db"}      # psnr_clean_db, ssim of the 2000-step point
    if rank == 0:
        # HBM traffic of the dominant kernel: PMC counters need rocprofv3, so this figure is NOT measured in this run; it is
        # read from the committed counter passes of this same command (tools/profile_round.sh: separate --pmc FETCH_SIZE /
        # WRITE_SIZE passes, 2*FETCH_SIZE + WRITE_SIZE per launch, the guide's gfx950 correction) and labelled as such
        traffic, traffic_src = None, None
        for tag in ("r05", "r04", "r03", "r02", "r01"):
            try:
                with open(os.path.join(ROOT, "profiles", tag + "_traffic.json")) as f:
                    traffic = float(json.load(f)["k_fused"]["hbm_bytes"])
                traffic_src = "profiles/%s_traffic.json (static: rocprofv3 --pmc passes of this command, not this run)" % tag
                break
            except Exception:
                pass
        train_f, fused_f, _ = flops_per_sample(LAYERS, FEATURES)
        peak = PEAKS[args.precision]
        if args.precision != "fp32" or args.config != "c2":
            traffic, traffic_src = None, None       # the committed counter passes are for the default configuration only
        fused_ms = tot_ms.value / max(launches.value, 1)
        achieved = fused_f * SAMPLE / (fused_ms * 1e-3) / 1e12
        ms_step = elapsed * 1e3 / args.steps
        value = SAMPLE * args.steps * world / elapsed
        out = {
            "metric": "encode_voxels_per_sec", "value": value, "unit": "voxels/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": DTYPES[args.precision], "data": "synthetic",
            "config": {"workload": "SingleTask 512^3 synthetic uint16 volume%s, SIREN %dx%d (layers=%d, features=%d, w0=20), "
                                   "randompoint sample_size=100000, datal2, Adamax lr=1e-3" %
                                   ("" if world == 1 else " per rank (DivideTask: %d independent 512^3 blocks)" % world,
                                    LAYERS - 1, FEATURES, LAYERS, FEATURES),
                       "volume": list(BLOCK), "layers": LAYERS, "features": FEATURES, "sample_size": SAMPLE,
                       "params": net.param_count, "bits_per_voxel": 32.0 * net.param_count / float(np.prod(BLOCK))},
            "roofline": {"bound": "mfma", "kernel": fused_kernel_name(args.precision, FEATURES)
                         + " (forward+loss+dgrad)", "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel_ms": fused_ms, "flop_per_launch": fused_f * SAMPLE,
                         "step_tflops": train_f * SAMPLE / (ms_step * 1e-3) / 1e12,
                         "step_frac": train_f * SAMPLE / (ms_step * 1e-3) / 1e12 / peak},
            "loss": float(loss.item()), "preroll_steps": pre,
            "psnr_at_bitrate": {"steps": steps_done, "bits_per_voxel": 32.0 * net.param_count / float(np.prod(BLOCK)), "psnr_db": psnr, **q2k_out},
        }
        out["ranks"] = gather_ranks(dist, rank, world, dev, backend)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        # the secondary numbers as flat scalars, EARLY in the line (`summary`) and once more inside `roofline` (which the driver's record keeps):
        # per-config ms_per_step / step_frac / kernel_frac, encode / decode wall clocks, the PSNR points
        sm = summarize({**out, **extra})
        out["summary"] = sm
        out["roofline"].update({"cfg." + k: v for k, v in sm.items() if k.endswith((".step_frac", ".kernel_frac"))})
        head = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "summary",
                "roofline", "cpu_baseline")
        out = {**{k: out[k] for k in head if k in out}, **{k: v for k, v in out.items() if k not in head}, **extra}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
