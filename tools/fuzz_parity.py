"""Randomised differential test of the C-ABI against the CPU oracle: random net shapes (layers, width, cin, cout, output_act, w0),
batch sizes (1 ... a few thousand, ragged), loss kinds, weight maps and thresholds; forward, loss and every gradient tensor.
    python tools/fuzz_parity.py [cases] [seed] [precision]      (GPU box; prints the failing configurations, exit code 1 if any)
precision bf16x3: widths <= 256; the forward under test is the TRAIN kernel's yhat AND m.forward() (inference runs k_fused_x3<false>, the fp16-halves chains)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from brief_pytorch_amd.networks import SIREN
from oracle import oracle as O

def relerr(a, b, floor=0.0):
    # floor: a batch of one or two samples whose outputs happen to sit near zero has no max|y| of its own to be relative to
    # (sine head, n = 1, y = -4.6e-4: an absolute 1.5e-6 read as 3e-3); the forward band is then taken against 1 % of the output range
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (max(np.max(np.abs(b)), floor) + 1e-30))

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
prec = sys.argv[3] if len(sys.argv) > 3 else 'fp32'
detail = int(sys.argv[4]) if len(sys.argv) > 4 else -1      # print every tensor's distances for this case
worst = [0.0, 0.0, 0.0]
plain = [0.0, 0.0]      # worst forward / gradient distance against the PLAIN bands
over = [0, 0, 0]        # cases outside the plain forward band | outside the plain gradient band | with a widened band
widths = list(range(1, 65)) + [65, 95, 96, 97, 127, 128, 129, 160, 191, 192, 200, 223, 224, 255, 256, 257, 288, 289, 300, 320, 383, 384, 385, 416, 450, 480, 511, 512,
          513, 527, 544, 545, 576, 600, 640, 641, 700, 768, 769, 800, 832, 896, 897, 960, 1000, 1023, 1024,      # round 4: every tile count 9 .. 32 (k_lean)
          1025, 1100, 1280, 1494, 2048]      # round 5: k_wide (passes of 5 .. 8 tile slots)
bad = 0
for case in range(cases):
    L = int(rng.integers(2, 12))
    F = int(rng.choice(widths))
    if prec != 'fp32' and F > 256:
        F = int(rng.choice([96, 128, 200, 256]))
    if F > 256 and L > 6:
        L = int(rng.integers(2, 7))                      # keep the oracle quick
    if F > 512 and L > 4:
        L = int(rng.integers(2, 5))
    cin, cout = int(rng.choice([2, 3])), int(rng.choice([1, 1, 1, 2, 3, 4]))
    oa = bool(rng.random() < 0.15)
    w0 = float(rng.choice([10.0, 20.0, 30.0]))
    n = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 255, 257, 1000, 2049, int(rng.integers(1, 6000))]))
    kind = int(rng.integers(0, 2))
    thr = float(rng.choice([0.0, 30.0, 200.0]))
    beta = float(rng.choice([0.01, 1.0, 20.0]))
    use_w = bool(rng.random() < 0.6)
    torch.manual_seed(case)
    m = SIREN(coords_channel=cin, data_channel=cout, features=F, layers=L, w0=w0, output_act=oa, precision=prec)
    d = O.make_desc(cin, cout, L, F, w0, 30.0, oa)
    p = m.params.numpy().copy()
    m.to('cuda')
    x = rng.uniform(-1, 1, size=(n, cin)).astype(np.float32)
    y = (rng.uniform(-1, 1, size=(n, cout)) if oa else rng.uniform(0, 100, size=(n, cout))).astype(np.float32)
    w = np.where(rng.uniform(size=(n, cout)) < 0.5, 0.25, 1.0).astype(np.float32) if use_w else np.ones((n, cout), np.float32)
    tag = "case %d: L=%d F=%d cin=%d cout=%d oa=%d w0=%g n=%d loss=%d thr=%g beta=%g weights=%d" % (case, L, F, cin, cout, oa, w0, n, kind, thr, beta, use_w)
    try:
        yh = m.forward(torch.from_numpy(x).cuda()).cpu().numpy()
        yo = O.forward(d, p, x)
        e_f = relerr(yh, yo, 0.01 * (1.0 if oa else 100.0))
        # conditioning of the forward (a deep narrow net under a sine head multiplies every rounding error by w0 per layer): the band is 2e-5
        # or three times the oracle's own f32 <-> f64 distance, whichever is larger
        f_band = max(2e-5, 3.0 * relerr(yo, O.forward(d, p, x, f64=True), 0.01 * (1.0 if oa else 100.0)))
        loss, yt = m.train_step(n, torch.from_numpy(y).cuda(), coords=torch.from_numpy(x).cuda(), weights=torch.from_numpy(w).cuda() if use_w else None,
                                loss=["datal2", "datasmoothl1"][kind], thr=thr, beta=beta, want_yhat=prec != 'fp32')
        if prec != 'fp32':      # both forwards are fuzzed: the inference kernel's (e_f above) and the TRAIN kernel's yhat
            e_f = max(e_f, relerr(yt.cpu().numpy(), O.forward(d, p, x), 0.01 * (1.0 if oa else 100.0)))
        lo, go, yho, _ = O.loss_grad(d, p, x, y, w, kind, thr, beta)
        _, go64, _, _ = O.loss_grad(d, p, x, y, w, kind, thr, beta, f64=True)
        e_l = abs(loss.item() - lo) / (abs(lo) + 1e-30)
        gw, gb = O.unpack_params(d, go)
        gw64, gb64 = O.unpack_params(d, go64)
        # conditioning: how far the oracle's own f32 gradient is from its f64 one (deep one- to five-wide nets multiply every
        # rounding error by w0 per layer); the band is 1e-4 or three times that, whichever is larger
        cond = max(max(relerr(gw[l], gw64[l]), relerr(gb[l], gb64[l])) for l in range(L))
        g_band = max(1e-4, 3.0 * cond)
        if kind == 1:
            # smooth-L1 has a kink at |yhat - y| = beta (slope df / beta inside, +-1 outside meet there, but a sample that sits within the forward's
            # rounding of it and lands on different sides in the two implementations changes its own term by up to that rounding / beta): the band
            # widens by one sample's share for every sample that close
            near = np.abs(np.abs(np.asarray(yho, np.float64) - y) - beta) < 1e-5 * np.maximum(1.0, np.abs(y))
            g_band += float(np.count_nonzero(near)) * 2.0 / max(1, n * cout)
        mw, mb = O.unpack_params(d, m.grads.cpu().numpy())
        gmax = max(float(np.max(np.abs(t))) for t in list(gw) + list(gb))
        if case == detail:
            for l in range(L):
                print("  layer %d: W hip-o32 %.2e  o32-o64 %.2e  max|g| %.2e   b hip-o32 %.2e  o32-o64 %.2e  max|g| %.2e" % (
                    l, relerr(mw[l], gw[l]), relerr(gw[l], gw64[l]), float(np.max(np.abs(gw[l]))), relerr(mb[l], gb[l]), relerr(gb[l], gb64[l]), float(np.max(np.abs(gb[l])))))
                if F <= 2:
                    print("    W hip", np.asarray(mw[l]).ravel()[:4], "o32", np.asarray(gw[l]).ravel()[:4], "o64", np.asarray(gw64[l]).ravel()[:4])
        e_g = 0.0
        for l in range(L):
            for a, b in ((mw[l], gw[l]), (mb[l], gb[l])):
                # a tensor whose gradient is (numerically) zero everywhere has no max-abs of its own to be relative to
                # ... and a tensor whose gradient nearly cancels over the batch (case 523 of seed 101: a bias gradient of 7e-4 summed from 128 terms of
                # ~0.5 beside gradients of 1.9) carries the ~1e-6 absolute error of v_sin / v_cos in every term: an absolute 3e-6 there is the f32
                # level of the NET's gradients, not 4e-3 of anything.  Hence a floor of 5e-2 of the largest gradient under the 1e-4 band (5e-6 absolute).
                scale = max(float(np.max(np.abs(b))), 5e-2 * gmax, 1e-30)
                e_g = max(e_g, float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))) / scale))
        # (a loss that is a small difference of O(1) numbers - one sample, sine head - is only as exact as yhat is: its band widens with the forward's)
        ok = e_f < f_band and abs(loss.item() - lo) <= 1e-5 * (f_band / 2e-5) * max(abs(lo), 1e-2 * float(np.mean(y.astype(np.float64) ** 2))) and e_g < g_band and np.isfinite(loss.item())
    except Exception as ex:
        ok, e_f, e_l, e_g = False, -1, -1, -1
        tag += "  EXCEPTION %r" % (ex,)
    if e_f >= 0 and e_g < 1.0:
        worst = [max(worst[0], e_f / (f_band / 2e-5)), max(worst[1], e_l), max(worst[2], e_g / (g_band / 1e-4))]
        plain = [max(plain[0], e_f), max(plain[1], e_g)]
        over = [over[0] + (e_f >= 2e-5), over[1] + (e_g >= 1e-4), over[2] + (f_band > 2e-5 or g_band > 1e-4)]
    if not ok:
        bad += 1
        print("FAIL %s  forward %.2e loss %.2e grads %.2e" % (tag, e_f, e_l, e_g), flush=True)
    elif case % 25 == 0:
        print("ok   %s  forward %.1e loss %.1e grads %.1e" % (tag, e_f, e_l, e_g), flush=True)
# the PLAIN bands first (forward 2e-5, gradients 1e-4, no scaling), then the conditioning-scaled ones the pass / fail decision uses
print("%d cases; PLAIN bands: worst forward %.2e (band 2e-5: %d cases outside), worst gradient tensor %.2e (band 1e-4: %d cases outside); %d cases had a band widened by the oracle's own f32 <-> f64 distance" %
      (cases, plain[0], over[0], plain[1], over[1], over[2]))
print("%d cases, %d failures; worst forward %.2e of a 2e-5 band (conditioning-scaled), loss %.2e, gradients %.2e of a 1e-4 band (conditioning-scaled)" % (cases, bad, worst[0], worst[1], worst[2]))
sys.exit(1 if bad else 0)
