// brief_hip.hip — gfx950 (MI355X, CDNA4) kernels + C-ABI for BRIEF's SIREN fit/decode path.
//
// Reference path replaced (citations into RichealYoung/BRIEF_PyTorch):
//   SIREN.forward            utils/Networks.py:269-271, Sine :227-234
//   loss + autograd backward main.py:176-191, 391-396
//   optimizer step           utils/misc.py:174-183 (torch.optim.Adamax/Adam/SGD), main.py:399
//   samplers / coords        main.py:126-163, utils/dataset.py:11-62
//   decode + de-normalise    main.py:266-297, utils/misc.py:59-92, utils/io.py:136-147
//
// Design (DESIGN.md has the full story).  Everything is computed in the TRANSPOSED form
//   Z^T[feature][sample] = W[feature][k] * H^T[k][sample]
// with v_mfma_f32_32x32x2_f32 (exact f32, 64 cycles/SIMD).  In that form the C/D accumulator
// layout (lane&31 = sample, register = feature row) is exactly the B-operand layout of the next
// layer's MFMA, so activations move between layers as a plain "register image" (no transposes):
// each wave writes its 32-feature tiles to LDS and every wave of the workgroup reads the whole
// image back as B operands.  The weights are the A operand; brief_siren_repack stores them in
// fragment order so each wave-instruction loads one contiguous 1 KiB block (L2 resident).
//
// One translation unit: this file holds the C-ABI and the launch logic, the kernels live in the *.inc files included below
// (brief_device.inc shared definitions, brief_fused.inc k_fused, brief_lean.inc k_lean, brief_small.inc k_small, brief_wgrad.inc k_wgrad,
//  brief_x3.inc split precision, brief_reduce.inc optimizer + k_reduce, brief_bf16.inc the bf16 path, brief_aux.inc repack / metrics / deblocking).
//
//   k_fused<NT,TRAIN>  coords -> layer0 -> hidden layers -> head -> loss -> dgrad chain.
//                      TRAIN stores Z_l (phases w z reduced to revolutions) and D_l (deltas) as [feature][sample]
//                      panels for the weight-gradient GEMM and accumulates the skinny gradients
//                      (first layer, head) itself.
//   k_wgrad<NT>        dW_l = D_l * sin(2 pi Z_{l-1})^T, split-K over sample chunks, fp32 slabs.
//   k_small<NT,HB>     F <= 64 (what BRIEF's YAMLs produce): the whole train step without any HBM stash, z in
//                      registers, dW accumulated in registers across the persistent tile loop.
//   k_reduce           deterministic slab/record reduction -> canonical gradient buffer + loss
//                      (+ fused optimizer and write-through to the fragment copies in brief_siren_fit_step).
//   k_optim            Adamax / Adam / SGD elementwise update (bit-matches the oracle's rule).
//   k_repack           canonical params -> fragment-ordered copies (fp32, and bf16 when precision = BF16).
//   brief_bf16.inc     k16 / k_wgrad16 / k_reduce16: the same path on v_mfma_f32_32x32x16_bf16 (BRIEF_PREC_BF16).
//   k_sample, k_sse_u16, k_ssim_u16, k_deblock_edge: index stream, metrics and the deblocking filter.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <mutex>
#include "brief_layout.h"
#include "brief_math.h"

#include "brief_device.inc"     // shared device-side definitions
#include "brief_fused.inc"      // k_fused
#include "brief_lean.inc"       // k_lean (run-time widths)
#include "brief_wide.inc"       // k_wide (widths above 1024 features)
#include "brief_small.inc"      // k_small, k_small_group
#include "brief_wgrad.inc"      // k_wgrad
#include "brief_x3.inc"         // split precision: k_fused_x3, k_wgrad_x3
#include "brief_reduce.inc"     // optimizer, k_reduce, k_reduce_group
#include "brief_bf16.inc"       // bf16: k16, k_wgrad16, k_reduce16
#include "brief_aux.inc"        // k_repack, index stream, metrics, deblocking filter

// =============================================================================================
// C-ABI
static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, const char *detail = "")
{
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}
#define HIP_TRY(x)                                                                  \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) return fail(BRIEF_ERR_LAUNCH, #x ": %s", hipGetErrorString(e_)); \
    } while (0)

static int check_desc(const brief_siren_desc *d)
{
    if (!d) return fail(BRIEF_ERR_INVALID, "null desc");
    if (d->cin != 2 && d->cin != 3) return fail(BRIEF_ERR_INVALID, "coords_channel must be 2 or 3");
    if (d->cout < 1 || d->cout > 4) return fail(BRIEF_ERR_INVALID, "data_channel must be 1..4");
    if (d->layers < 2) return fail(BRIEF_ERR_INVALID, "layers must be >= 2");
    if (d->features < 1 || d->features > 32 * BRIEF_MAX_NT)
        return fail(BRIEF_ERR_INVALID, "features must be 1..4096 on the fused path");
    if (d->precision == BRIEF_PREC_BF16 && d->features > 512)
        return fail(BRIEF_ERR_INVALID, "BRIEF_PREC_BF16 supports features <= 512 (wider nets run in BRIEF_PREC_F32)");
    if (d->precision != BRIEF_PREC_F32 && d->precision != BRIEF_PREC_BF16 && d->precision != BRIEF_PREC_BF16X3)
        return fail(BRIEF_ERR_INVALID, "precision must be BRIEF_PREC_F32, BRIEF_PREC_BF16 or BRIEF_PREC_BF16X3");
    if (d->precision == BRIEF_PREC_BF16X3 && d->features > 256) return fail(BRIEF_ERR_INVALID, "BRIEF_PREC_BF16X3 supports features <= 256");
    return 0;
}

// optional live timing of the dominant kernel (bench.py roofline leg): event pairs recorded on the
// caller's stream around every k_fused<TRAIN> launch while enabled.
static const int kProfSlots = 4096;
static bool g_prof_on = false;
static int g_prof_n = 0;
static int g_prof_dev = -1;             // the device brief_profile_enable was called on: only its launches are timed

// ---- per-device library state.  One process per GPU is the deployment, but nothing here may silently assume it: the LDS-size
// attributes of the kernels, the stream / event pool of brief_multi_fit and the timing events all belong to ONE device, so they
// are keyed by hipGetDevice() (a process that touches a second device gets its own attributes, streams and events there).
static const int kMaxDevices = 64;
static const int kPoolStreams = 8;
struct DevState {
    bool pool_init, prof_init;
    hipStream_t pool[kPoolStreams];
    hipEvent_t pool_ev[kPoolStreams + 1];
    bool aux_init;                      // the side stream of the bf16 overlap plan (train_step_impl) and its fork / join events
    hipStream_t aux;
    hipEvent_t aux_fork, aux_join;
    bool aux2_init;                     // ... and of the fp32 tail plan (fused_tail_plan)
    hipStream_t aux2;
    hipEvent_t aux2_fork, aux2_join;
    hipEvent_t *prof_ev;                // [2 * kProfSlots]
    int nattr;
    const void *attr_fn[256];           // kernels whose hipFuncAttributeMaxDynamicSharedMemorySize was set on this device
    int attr_bytes[256];
};
static int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    return dev;
}
// Host threads: the per-device state (attributes set, stream / event pools) is created lazily, and a train step that forks onto a side stream records and waits on the
// device's ONE pair of fork / join events — two threads enqueueing on one device at the same time would wait on each other's records.  Everything that touches that state
// runs under this (recursive: brief_multi_fit -> fit steps) lock; the calls only ENQUEUE, so it is held for microseconds.  Different devices do not share state but share
// the lock (one process per GPU is the deployment: SURVEY 8e).
static std::recursive_mutex g_state_mu;
static DevState *dev_state()
{
    std::lock_guard<std::recursive_mutex> lk(g_state_mu);
    static DevState *tab[kMaxDevices] = {};
    const int dev = current_device();
    if (!tab[dev]) tab[dev] = (DevState *)calloc(1, sizeof(DevState));
    return tab[dev];
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel, size)
static int dev_attr_once(const void *fn, int bytes)
{
    std::lock_guard<std::recursive_mutex> lk(g_state_mu);
    DevState *ds = dev_state();
    if (!ds) return fail(BRIEF_ERR_INVALID, "out of host memory");
    for (int i = 0; i < ds->nattr; ++i)
        if (ds->attr_fn[i] == fn && ds->attr_bytes[i] >= bytes) return 0;
    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (ds->nattr < 256) { ds->attr_fn[ds->nattr] = fn; ds->attr_bytes[ds->nattr] = bytes; ++ds->nattr; }
    return 0;
}

// compute units of the device the calling thread is on (256 on a whole MI355X; fewer on a partitioned one).  Grids,
// workspace layout and the start stagger are sized from it, queried once per device.
static int cu_count()
{
    static int cached[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}
#define kCUs cu_count()
#define kWgradBlocks cu_count()

// persistent grid: g_wg_per_cu workgroups per CU, each walking tiles blockIdx, blockIdx+grid, ...
// scalars prepared in double exactly as oracle_optim_step / torch do
static OptimScalars optim_scalars(int kind, double lr, double beta1, double beta2, double eps, int64_t t)
{
    OptimScalars o;
    o.kind = kind;
    o.w1 = (float)(1.0 - beta1); o.fb2 = (float)beta2; o.w2 = (float)(1.0 - beta2); o.feps = (float)eps;
    o.nstep = kind == BRIEF_OPT_SGD ? (float)(-lr) : (float)(-(lr / (1.0 - pow(beta1, (double)t))));
    o.bc2s = kind == BRIEF_OPT_ADAM ? (float)sqrt(1.0 - pow(beta2, (double)t)) : 1.f;
    return o;
}

// Diagnostic knobs (BRIEF_WG_PER_CU, BRIEF_STAGGER, BRIEF_DIAG, BRIEF_TAIL_ROUNDS, ...): read from the environment once, and ONLY in a
// -DBRIEF_DIAGNOSTICS build (tools/ build their own library and point BRIEF_LIB at it).  The product library never calls getenv: no
// environment variable can change what a fit computes or how it is launched (tests/test_host_logic.py checks the symbol table).
#ifdef BRIEF_DIAGNOSTICS
static const char *env_str(const char *name) { return getenv(name); }
#else
static const char *env_str(const char *) { return nullptr; }
#endif
static int env_int(const char *name, int dflt, int lo, int hi)
{
    const char *e = env_str(name);
    if (!e) return dflt;
    const int v = atoi(e);
    return v >= lo && v <= hi ? v : dflt;
}
static const int kRecWgsPerCu = 8;      // per-workgroup record slots per CU in the workspace (k_fused: body + single-tile tail)
static const int g_wg_per_cu = env_int("BRIEF_WG_PER_CU", BRIEF_TRAIN_WPE, 1, 4);
static const bool g_wg_per_cu_set = env_str("BRIEF_WG_PER_CU") != nullptr;
static const int g_stagger = env_int("BRIEF_STAGGER", 0, 0, 64);      // start delay per residency slot: measured neutral with two and with three workgroups per CU (profiles/r03_wg_timeline.md), off
static const int g_diag = env_int("BRIEF_DIAG", 0, 0, 255);
static const int g_x3_decode = env_int("BRIEF_X3_DECODE", 1, 0, 1);      // 0 (diagnostics): BRIEF_PREC_BF16X3 nets are evaluated by the f32 forward kernel
static const int g_wgrad_repeat = env_int("BRIEF_WGRAD_REPEAT", 1, 1, 8);      // diagnostics: k_wgrad_x3 launched this many times per step (reads of data that k_fused has just written vs data at rest)
static const int g_reduce_sg_big = env_int("BRIEF_REDUCE_SG_BIG", 4, 1, 64);
static const int g_reduce_part = env_int("BRIEF_REDUCE_PART", 0, 0, 2);        // diagnostics: 1 / 2 = launch only k_reduce's hidden-parameter / skinny blocks (what each costs alone; results are then incomplete)
static const int g_reduce_vec = env_int("BRIEF_REDUCE_VEC", 1, 0, 1);          // diagnostics: 0 = one hidden parameter per k_reduce thread everywhere
static const int g_reduce_sg_small = env_int("BRIEF_REDUCE_SG", 0, 0, 64);      // diagnostics: k_reduce threads per hidden parameter behind k_small (power of two; 0 = by slab count)
// samples one workgroup tile covers: 32 per sample sub-tile; the split-precision TRAIN kernel walks 64-sample tiles
static int64_t fused_wg_samples(const brief_siren_desc &d, bool train)
{
    if (brief_use_wide(d)) return 32;                                                      // k_wide: one 32-sample tile
    if (d.precision == BRIEF_PREC_F32 && brief_use_lean(d, train)) return 32;              // k_lean<1, ...>: one 32-sample tile
    if (!train && d.precision == BRIEF_PREC_F32) return 32 * (4 / brief_wm_infer(brief_nt(d)));      // KCfg<NT, true>
    return brief_wg_samples(brief_nt(d));      // (the split-precision TRAIN kernel deals 32-sample half-tiles too, and walks them in pairs)
}
static int fused_grid(const brief_siren_desc &d, int64_t n, bool train)
{
    const int nt = brief_nt(d);
    const int64_t wgs_ = fused_wg_samples(d, train);
    const int64_t tiles = (n + wgs_ - 1) / wgs_;
    // resident workgroups per CU = what the kernel's launch bounds were compiled for (BRIEF_WG_PER_CU: diagnostics)
    const int wpe = g_wg_per_cu_set ? g_wg_per_cu : (d.precision == BRIEF_PREC_BF16X3 && (train || g_x3_decode) ? 2 /* k_fused_x3 */ :
                    (BRIEF_FUSED64 && train && nt == 8 ? (BRIEF_FUSED64 == 1 ? 2 : 3) /* k_lean<2, 2, 8> / k_lean<1, 2, 8> */ : (train ? fused_train_wpe(nt) : (nt > 8 ? 2 : 3))));
    int64_t cap = (int64_t)kCUs * (train && nt > 8 ? 1 : wpe);      // TRAIN > 8 tiles: 512-register kernel, one workgroup per CU
    if (brief_use_wide(d)) cap = kCUs;                               // k_wide: a 128 KB slab per workgroup
    if (d.precision == BRIEF_PREC_F32 && brief_use_lean(d, train))      // k_lean: what its launch bounds and its LDS image allow
    {
        const int64_t by_lds = (160 * 1024) / (int64_t)(sizeof(float) * lean_lds(1, (nt + 3) / 4, nt).total);
        const int64_t by_regs = lean_wpe(1, (nt + 3) / 4);
        cap = (int64_t)kCUs * (by_lds < by_regs ? (by_lds > 0 ? by_lds : 1) : by_regs);
    }
    return (int)(tiles < cap ? (tiles > 0 ? tiles : 1) : cap);
}
// k_fused<TRAIN> launch plan: a persistent body of `cap` workgroups over whole rounds of tiles, the rest of the batch
// (the last BRIEF_TAIL_ROUNDS full rounds and the remainder) as single-tile workgroups behind them in the same grid
struct FusedPlan { int grid, pers_wgs; int64_t pers_tiles; };
static const int g_tail_rounds = env_int("BRIEF_TAIL_ROUNDS", 0, 0, 64);
static FusedPlan fused_plan(const brief_siren_desc &d, int64_t n, bool train)
{
    const int nt = brief_nt(d);
    const int64_t wgs_ = fused_wg_samples(d, train);
    const int64_t tiles = (n + wgs_ - 1) / wgs_;
    const int cap = fused_grid(d, n, train);
    FusedPlan p;
    p.grid = cap; p.pers_wgs = cap; p.pers_tiles = tiles;
    if (!train || nt > 8 || tiles <= cap || brief_use_wide(d)) return p;        // decode: millions of tiles, the tail does not matter; > 8 tiles: one workgroup per CU
    if (g_tail_rounds == 0) return p;
    int64_t rounds = tiles / cap - g_tail_rounds;
    if (rounds < 0) rounds = 0;
    const int64_t max_wgs = (int64_t)kCUs * kRecWgsPerCu;               // record slots (ws_layout)
    while (tiles - rounds * cap + (rounds ? cap : 0) > max_wgs) ++rounds;
    if (rounds * cap >= tiles) return p;
    p.pers_wgs = rounds ? cap : 0;
    p.pers_tiles = rounds * cap;
    p.grid = p.pers_wgs + (int)(tiles - p.pers_tiles);
    return p;
}
#ifndef BRIEF_WGRAD_ROUNDS
#define BRIEF_WGRAD_ROUNDS 1
#endif
#ifndef BRIEF_TAIL_PLAN
#define BRIEF_TAIL_PLAN 1
#endif
#ifndef BRIEF_TAIL_LEAN
#define BRIEF_TAIL_LEAN 0
#endif
#ifndef BRIEF_TAIL_MODE_DEFAULT
#define BRIEF_TAIL_MODE_DEFAULT 2
#endif
#ifndef BRIEF_TAIL_R
#define BRIEF_TAIL_R 1.15      // measured (tools/width_sweep.py over r = 0.5 .. 2.0, diagnostics build): 1.1 .. 1.2 is the flat optimum at 320, 384, 448 and 512 features
#define BRIEF_TAIL_R1 1.05     // ... above 16 tiles (one k_lean workgroup per CU: the body's tiles run solo too): 0.9 / 1.05 / 1.25 -> 4x768 8.215 / 8.179 / 8.220 ms
#endif
#ifndef BRIEF_TAIL_MIN_SPLITS
#define BRIEF_TAIL_MIN_SPLITS 6
#endif
#ifndef BRIEF_TAIL_MIN_ROUNDS
#define BRIEF_TAIL_MIN_ROUNDS 3
#endif
// The tail plan of the wide fp32 nets (9 tiles and more: k_lean with two or one workgroup per CU, k_wide).  A 100 000-sample batch is 3 125 tiles; with G resident
// workgroups the last round of the fused kernel holds 3 125 mod G tiles (53 for G = 256 or 512) and the other CUs idle for a whole tile time — 1 / 12.2 of the
// launch at one workgroup per CU, i.e. ~6 % of it.  Nothing in k_wgrad's work on the BODY's chunks depends on those tiles, so:
//     stream:        fused body (whole rounds) ─┬─ fused tail (the last tiles) ─ k_wgrad over the tail's chunks (one more K split) ─┬─ k_reduce
//     side stream:                              └─ k_wgrad over the body's chunks ───────────────────────────────────────────────────┘
// k_wgrad's workgroups fill the CUs the tail leaves free and, being several rounds of short workgroups, take over the tail's CUs as they become free.  (At 8 tiles —
// the headline — this loses: its k_wgrad is ONE round of 255 one-per-CU workgroups, and the slabs three rounds would need cost k_reduce more than the tail: DESIGN.md.)
// Slabs are summed in slab order; the plan is a function of the job alone, so results stay bit-reproducible.
// Measured (tools/ab_step.sh, three interleaved rounds, profiles/r05_tail_plan.md): 4x1494 33.85 -> 32.68 ms (+3.5 %), 4x1100 19.63 -> 19.45 (+0.9 %); where the body's
// k_wgrad has FEW long workgroups the end of the step is quantised by its rounds and the plan loses (4x1800 52.16 -> 52.65, 4x2048 59.45 -> 59.70: 4 splits of 192),
// and k_lean's nets (9 .. 32 tiles) gain nothing at their split counts (4x1024 14.58 -> 14.62, 4x512 3.90 -> 4.15 with the 64 splits three rounds need): the plan is
// taken for k_wide when the body splits at least 6 ways.
//
// UNEVEN form (mode 2).  The form above adds a K split and needs k_wgrad in many short workgroups.  What the late CUs really need is LESS WORK, not finer work: k_wgrad's
// workgroups go to CUs in index order, the `rem` CUs the tail holds for T take the LAST ones — so the last n_short K splits are made shorter by the T they start late,
//     k_norm = (chunks + n_short rho) / nsplit,   k_short = k_norm - rho,   rho = T in units of one workgroup's time per 32-sample chunk,
// and their workgroups numbered last (split-major order).  The short splits also take the tail's own chunks (the end of the chunk range), so they are a launch of their own behind
// the tail on the caller's stream while the normal splits start on the side stream behind the body: same slab count, same k_reduce, every CU ends together:
//     stream:        fused body ─┬─ fused tail (rem workgroups, T) ─ k_wgrad, the n_short short splits ─┬─ k_reduce
//     side stream:               └─ k_wgrad, the nsplit - n_short normal splits ───────────────────────┘
// rho = r 2 (L - 2) nt^2 / QT^2 (a tile's forward + dgrad chains against one chunk of a QT x QT quadrant, both on four SIMDs), r = how much slower a SOLO tile runs than
// k_wgrad's steady rate (BRIEF_TAIL_R, measured).
struct TailPlan { int mode /* 0 off, 1 extra split, 2 uneven splits */; bool on; int grid_body, grid_tail; int64_t tiles_body; int ns, n_norm; int64_t kfix_n, kfix_s; };
static int wgrad_split_rule(const brief_siren_desc &d, int64_t nchunks, int min_rounds);
static const int g_tail_mode = env_int("BRIEF_TAIL_MODE", BRIEF_TAIL_MODE_DEFAULT, 0, 2);      // diagnostics: 0 no tail plan, 1 the extra-split form only, 2 uneven splits where eligible
static const int g_tail_ntmax = env_int("BRIEF_TAIL_NTMAX", 32, 9, 32);                        // diagnostics: the uneven form up to this many tiles
static const int g_tail_r = env_int("BRIEF_TAIL_R", 0, 0, 1000);                               // diagnostics: r in percent (0: the class's own)
static TailPlan fused_tail_plan(const brief_siren_desc &d, int64_t n)
{
    TailPlan p;
    memset(&p, 0, sizeof(p));
    if (!BRIEF_TAIL_PLAN || g_tail_mode == 0 || d.precision != BRIEF_PREC_F32 || d.layers - 2 < 1) return p;
    const int nt = brief_nt(d);
    if (nt < 8) return p;
    const int64_t tiles = (n + 31) / 32;
    const int G = fused_grid(d, n, true);
    const int64_t full = tiles / G * G, rem = tiles - full;
    if (full == 0 || rem == 0) return p;
    // (9 .. 32 tiles: k_lean, k_wgrad in ONE round.  9 .. 16 tiles, two workgroups per CU:  Measured, 100 000 samples, interleaved with the plan off on one box: 4x320 1.615 -> 1.587 ms,
    //  4x384 2.177 -> 2.143, 4x448 2.946 -> 2.859, 4x512 3.900 -> 3.794 (+1.8 / 1.6 / 3.0 / 2.8 %).  The headline's k_fused<8> gains 0.3 .. 0.7 % — a solo tail tile takes 70 us
    //  where the single launch's ragged end costs 41 — and stays one launch; where k_wgrad runs several rounds per XCD the uneven form loses to the even one
    //  (4x1024 with 14 + 2 splits 14.55 -> 14.97 ms, 4x1494 32.8 -> 34.7 against the extra-split form's 32.6).)
    if (g_tail_mode == 2 && nt >= 9 && nt <= g_tail_ntmax && rem * 4 <= 3 * (int64_t)kCUs) {
        const int hidden = d.layers - 2, nq = wgrad_nq(nt), qt = (nt + nq - 1) / nq;
        const int64_t chunks = brief_npad_d(d, n) / 32, B = (int64_t)hidden * nq * nq;
        int ns = wgrad_split_rule(d, chunks, 1);
        // 17 .. 32 tiles: the split rule's three rounds of k_wgrad (28 x 27 or 16 x 48 workgroups) are re-cut into ONE round (9 x 27 = 243, 5 x 48 = 240), where the uneven
        // form applies.  Measured against the three even rounds (tools/width_sweep.py, two interleaved rounds): 4x527 4.651 -> 4.524 ms, 4x640 6.235 -> 6.063, 4x768 8.464 -> 8.190
        // (+2.8 .. 3.3 %); 4x896 11.44 -> 11.39, 4x1024 14.60 -> 14.49 (+0.4 .. 0.8 %: five splits leave 16 CUs idle, which the even 16-split cut does not)
        if ((int64_t)ns * B > kCUs) ns = (int)(kCUs / B);
        // Workgroup i of a launch goes to XCD i mod 8, each XCD has CUs / 8 CUs and a k_wgrad workgroup needs a CU of its own: the normal splits' workgroups + the tail's, and
        // the normal + the short splits' workgroups, must fit every XCD's CUs, or one workgroup waits a whole round (measured: 201 + 53 on 256 CUs put 26 + 7 on XCD 0 —
        // k_wgrad 301 -> 400 + 492 us).  Only nets whose k_wgrad is ONE round are planned this way (8 .. 16 tiles).
        const int per_xcd = kCUs / 8, tail_x = (int)((rem + 7) / 8);
        int n_norm = 0;
        if ((int64_t)ns * B <= kCUs)
            for (int c = ns - 1; c >= 1; --c) {
                const int a_ = (int)((c * B + 7) / 8), s_ = (int)(((ns - c) * B + 7) / 8);
                if (a_ + tail_x <= per_xcd && a_ + s_ <= per_xcd && (ns - c) * B >= rem) { n_norm = c; break; }
            }
        const int n_short = ns - n_norm;
        const double r = g_tail_r ? 0.01 * g_tail_r : (nt <= 16 ? BRIEF_TAIL_R : BRIEF_TAIL_R1);
        const double rho = r * 2.0 * hidden * nt * nt / (double)(qt * qt);
        const double k_norm = ((double)chunks + n_short * rho) / ns, k_short = k_norm - rho;
        if (n_norm > 0 && k_short >= 4.0 && rho < 0.8 * k_norm) {
            p.kfix_n = (int64_t)(k_norm * 65536.0); p.kfix_s = (int64_t)(k_short * 65536.0);
            const int64_t b_norm = ((int64_t)(ns - n_short) * p.kfix_n) >> 16;      // first chunk of the short splits
            if (b_norm <= full && b_norm > 0) {                                      // (the normal splits stay inside the body's chunks)
                p.mode = 2; p.on = true; p.grid_body = G; p.grid_tail = (int)rem; p.tiles_body = full; p.ns = ns; p.n_norm = ns - n_short;
                return p;
            }
        }
    }
    // mode 1: the extra-split form, k_wide with many short k_wgrad workgroups
    if (!(brief_use_wide(d) || (BRIEF_TAIL_LEAN && brief_use_lean(d, true) && nt >= 9))) return p;
    if (rem * 4 > 3 * (int64_t)G) return p;      // (a last round that is three quarters full has nothing to give)
    if (wgrad_split_rule(d, full, BRIEF_TAIL_MIN_ROUNDS) < BRIEF_TAIL_MIN_SPLITS) return p;
    p.mode = 1; p.on = true; p.grid_body = G; p.grid_tail = (int)rem; p.tiles_body = full;
    return p;
}
// K splits of k_wgrad over `nchunks` 32-sample chunks.  min_rounds > 1 (the tail plan): at least that many rounds of quadrant workgroups
static int wgrad_split_rule(const brief_siren_desc &d, int64_t nchunks, int min_rounds)
{
    const int hidden = d.layers - 2;
    if (hidden <= 0) return 0;
    const int nq = wgrad_nq(brief_nt(d));
    const int64_t B = (int64_t)hidden * nq * nq;      // quadrant workgroups of one K split
    int64_t s = kWgradBlocks / B;
    if (s < 1) s = 1;
    if (min_rounds > 1) {
        double best = 1e30;
        for (int64_t c = 1; c <= 64; ++c) {
            const int64_t rounds = (c * B + kWgradBlocks - 1) / kWgradBlocks;
            if (rounds < min_rounds && c < 64) continue;
            const double cost = (double)rounds / (double)c + 0.0001 * (double)c;
            if (cost < best - 1e-9) { best = cost; s = c; }
        }
    } else if (brief_nt(d) > 32) {
        // above 1024 features one K split already has about as many quadrant blocks as the device has CUs (4x2048: 192, 4x1494: 108): take the
        // split count (<= 16) whose blocks fill whole rounds best.  A split more costs k_reduce one more slab per layer (FP^2 floats: 9 MB at 1 494
        // features = ~7 us at the 4 TB/s it reads) against a k_wgrad of ~12 ms: priced at 0.001 per split in this model — 4x1494 takes 7 splits
        // (756 workgroups = 2.95 rounds) where 2 (216 = 0.84 of a round) cost it 14 % of the launch
        double best = 1e30;
        for (int64_t c = 1; c <= 16; ++c) {
            const double cost = (double)((c * B + kWgradBlocks - 1) / kWgradBlocks) / (double)c + 0.001 * (double)c;
            if (cost < best - 1e-9) { best = cost; s = c; }
        }
    } else if (nq >= 2 && BRIEF_WGRAD_ROUNDS) {
        // run-time widths up to 1024 features: one round of s B <= CUs workgroups can leave CUs idle (25 .. 32 tiles, three hidden layers: 5 x 48 = 240 of
        // 256; eight sine layers of 1024: 2 x 112 = 224); a multiple-round split count is taken when it fills the rounds at least 1.5 % better.  A split more
        // costs k_reduce one slab per layer (FP^2 floats at ~4 TB/s: 1 us at 1024 features) against one quadrant workgroup's whole-K time (~25 ms):
        // 0.0001 of the unit this cost is counted in.  Measured (tools/ab_step.sh): 4x1024 14.89 -> 14.63 ms, 4x896 11.70 -> 11.48, 4x800 10.27 -> 10.08;
        // 17 .. 24 tiles (9 -> 28 splits = 756 workgroups = 2.95 rounds): 4x527 4.677 -> 4.649, 4x640 6.287 -> 6.242, 4x768 unchanged
        const double base = 1.0 / (double)s + 0.0001 * (double)s;
        double best = base;
        const int64_t cmax = 4 * s > 16 ? 4 * s : 16;
        for (int64_t c = s + 1; c <= cmax && c <= 64; ++c) {
            const double cost = (double)((c * B + kWgradBlocks - 1) / kWgradBlocks) / (double)c + 0.0001 * (double)c;
            if (cost < best - 1e-9 && cost < 0.985 * base) { best = cost; s = c; }
        }
    }
    if (s > nchunks) s = nchunks;
    return (int)s;
}
// total K splits (slabs per layer) of a train step: under the tail plan the body's splits + one for the tail's chunks
static int wgrad_splits(const brief_siren_desc &d, int64_t n)
{
    if (d.layers - 2 <= 0) return 0;
    const TailPlan tp = fused_tail_plan(d, n);
    if (tp.mode == 1) return wgrad_split_rule(d, tp.tiles_body, BRIEF_TAIL_MIN_ROUNDS) + 1;
    if (tp.mode == 2) return tp.ns;
    return wgrad_split_rule(d, brief_npad_d(d, n) / 32, 1);
}

// narrow nets (F <= 64, at most 7 hidden layers) train through k_small: no stash, no k_wgrad.
// BRIEF_SMALL=0 (diagnostics) sends them down the general k_fused + k_wgrad path instead.
static bool use_small(const brief_siren_desc &d)
{
    static int enabled = -1;
    if (enabled < 0) { const char *e = env_str("BRIEF_SMALL"); enabled = (e && atoi(e) == 0) ? 0 : 1; }
    return enabled && d.precision == BRIEF_PREC_F32 && brief_nt(d) <= 2 && d.layers - 2 <= 7;
}
static int small_hb(const brief_siren_desc &d)
{
    const int h = d.layers - 2;
    return h <= 1 ? 1 : (h <= 3 ? 3 : (h <= 5 ? 5 : 7));
}
// Workgroups (= gradient slabs) of a narrow-net step: a function of the job alone (never of what is trained beside it: a co-trained
// job uses the grid, tile walk and slab order of a fit on its own, bit for bit).  The tiles are spread evenly over the fewest rounds
// the resident capacity allows, but at least two per workgroup: a batch of up to 2 x capacity tiles takes the time of two tile
// passes with either count, and the many tiny blocks of a DivideTask — which share one k_small_group launch — then carry half
// the workgroup prologues, slab writes and k_reduce terms (round 4; it was min(tiles, capacity): one tile per workgroup for small jobs).
static int small_grid(const brief_siren_desc &d, int64_t n)
{
    const int nt = brief_nt(d);
    int64_t tiles = (n + brief_wg_samples(nt) - 1) / brief_wg_samples(nt);
    if (tiles < 1) tiles = 1;
    const int64_t cap = (int64_t)kCUs * small_wpe(small_hb(d));
    int64_t rounds = (tiles + cap - 1) / cap;
    if (rounds < 2) rounds = 2;
    return (int)((tiles + rounds - 1) / rounds);
}
// device table of a co-trained group (k_small_group / k_reduce_group): lives at the end of the FIRST job's workspace
static const int64_t kGroupTableFloats = (int64_t)BRIEF_GROUP_MAX * ((sizeof(FusedArgs) + sizeof(ReduceArgs) + 15) / 16 * 4 + 8);

// ---- bf16 path geometry: 128-sample workgroup tiles, one workgroup per CU; stashes in bf16 (2 per float slot)
static int64_t npad16(int64_t n) { return (n + 127) / 128 * 128; }
static int grid16(int64_t n)
{
    const int64_t tiles = (n + 127) / 128;
    return (int)(tiles < kCUs ? (tiles > 0 ? tiles : 1) : kCUs);
}
// How a batch is cut into launches of k16: a body of whole rounds of 128-sample tiles (one per CU), then what is left
// as quarter (NS = 1) or half (NS = 2) tiles when that spreads it over more CUs.  Multi-channel nets (CO = 4 kernels)
// only exist with NS = 4.
struct Split16 { int64_t n_body; int g_body, ns_tail, g_tail; };
static Split16 split16(const brief_siren_desc &d, int64_t n)
{
    Split16 s;
    const int64_t tiles = (n + 127) / 128;
    const int64_t full = tiles / kCUs * kCUs, rem = tiles - full;
    s.n_body = n; s.g_body = (int)(tiles < kCUs ? tiles : kCUs); s.ns_tail = 0; s.g_tail = 0;
    if (d.cout != 1 || rem == 0) return s;
    int ns = rem * 4 <= kCUs ? 1 : (rem * 2 <= kCUs ? 2 : 0);
    if (!ns) return s;
    s.n_body = full * 128;
    s.g_body = full > 0 ? kCUs : 0;
    s.ns_tail = ns;
    s.g_tail = (int)((npad16(n) - s.n_body) / (32 * ns));          // tiles up to the padded size (see k16)
    return s;
}
// The overlap plan of the 512-wide bf16 step (BASELINE config 3).  A 100 000-sample batch is 768 full 128-sample tiles + a remainder that runs as a
// second, quarter-tile launch of k16: 109 us for 1.7 % of the samples (a tile of any size streams the net's 7 MB of bf16 fragments through its
// CU), and the skinny first-layer / head job behind the weight-gradient GEMM is another 48 us of latency-bound launch — 157 us in which most of
// the chip idles.  Neither depends on k_wgrad16_big's work on the BODY's samples, so:
//     stream:      k16 body ─┬─ k_wgrad16_big over the body's sample blocks, na splits (hidden x 4 x na workgroups <= CUs - tail's share) ─┬─ k_reduce ...
//     side stream:           └─ k16 tail ─ k_wgrad16_big over the tail's blocks (one more split) ─ skinny job ───────────────────────────────┘
// The side work runs on the CUs the first launch leaves free (its workgroups own a CU each: 512 threads x 256 registers), in its shadow.
// Slabs are summed by k_reduce in slab order as before: results depend on the plan (a function of the job alone), not on timing.
struct Split16Plan { bool on; int na; int64_t kb_body; };      // na: K splits of the body part (the tail's blocks are split na)
static Split16Plan split16_plan(const brief_siren_desc &d, int64_t n)
{
    Split16Plan p;
    p.on = false; p.na = 0; p.kb_body = 0;
    const int hidden = d.layers - 2;
    if (d.precision != BRIEF_PREC_BF16 || brief_nt(d) != 16 || hidden < 1) return p;
    const Split16 sp = split16(d, n);
    if (sp.g_tail <= 0 || sp.g_body <= 0) return p;
    const int free_cus = (sp.g_tail + 1) / 2;                       // the tail's workgroups in at most two rounds
    int na = (kCUs - free_cus) / (hidden * 4);
    if (na > 15) na = 15;
    if (na < 1 || sp.n_body / 64 < na) return p;
    p.on = true; p.na = na; p.kb_body = sp.n_body / 64;
    return p;
}

static int wgrad16_overlap_splits(const brief_siren_desc &d, int64_t n)
{
    const Split16Plan p = split16_plan(d, n);
    return p.on ? p.na + 1 : 0;
}
static int wgrad16_splits(const brief_siren_desc &d, int64_t n)
{
    if (const int ov = wgrad16_overlap_splits(d, n)) return ov;      // the overlap plan: its body splits + one for the tail
    const int nt = brief_nt(d);
    const int hidden = d.layers - 2 > 0 ? d.layers - 2 : 0;
    const int64_t nblk = npad16(n) / 64;          // K is split in 64-sample blocks; every split needs one
    int64_t s;
    if (nt == 16) {
        // k_wgrad16_big: one 512-thread workgroup per CU; a whole number of workgroups per CU-round, not 1.1 rounds
        const int per = hidden * 4;               // (512 / 256)^2 output blocks per layer
        s = per > 0 ? kCUs / per : 1;
    } else {
        const int nb = nt / 4;                    // 128 x 128 output blocks per side
        const int per = hidden * nb * nb + 2 * nb;
        s = (2 * kCUs + per - 1) / per;
    }
    if (s < 1) s = 1;
    if (s > 16) s = 16;
    if (s > nblk) s = nblk;
    return (int)s;
}
static int wgrad16_skinny_splits(const brief_siren_desc &d, int64_t n)
{
    // the two skinny jobs (first layer, head) stream a whole stash plane each through only 2 * nb row blocks
    const int nb = brief_nt(d) / 4;
    int64_t s = (2 * kCUs) / (2 * nb);
    const int64_t nblk = npad16(n) / 64;
    if (s > nblk) s = nblk;
    return (int)(s < 1 ? 1 : s);
}
struct Ws16 { int64_t h, dd, x, g, rec, slabs, total; };      // offsets in floats (h: the fp16 phase planes, dd: the bf16 delta planes)
static Ws16 ws16_layout(const brief_siren_desc &d, int64_t n)
{
    const int64_t FP = 32 * brief_nt(d), np = npad16(n), planes = d.layers - 1, hidden = d.layers - 2 > 0 ? d.layers - 2 : 0;
    const int64_t stash = planes * FP * np / 2;                       // bf16 elements -> float slots (np is a multiple of 128)
    Ws16 w;
    w.h = 0; w.dd = w.h + stash;
    w.x = w.dd + stash; w.g = w.x + 4 * np / 2;
    w.rec = w.g + 4 * np / 2;
    w.slabs = w.rec + (int64_t)2 * kCUs * 8 /* body + tail launches */ + (int64_t)kCUs * 8 * 10 /* diagnostic stamps */;
    w.total = w.slabs + hidden * (FP * FP + FP) * (int64_t)wgrad16_splits(d, n) + 2 * FP * 4 * (int64_t)wgrad16_skinny_splits(d, n);
    return w;
}

struct WsLayout { int64_t z, dd, rec, slabs, table, img, total; };
// per-wave records of the fused kernel (k_wide: one workgroup per CU + the tail launch's; everything else: up to kRecWgsPerCu)
static int64_t rec_region_floats(const brief_siren_desc &d)
{
    return (int64_t)kCUs * (brief_use_wide(d) ? 2 : kRecWgsPerCu) * 4 * brief_rec_floats(brief_nt(d));
}
static WsLayout ws_layout(const brief_siren_desc &d, int64_t n)
{
    if (d.precision == BRIEF_PREC_BF16) {
        WsLayout w; w.z = w.dd = w.rec = w.slabs = w.table = w.img = 0; w.total = ws16_layout(d, n).total;
        return w;
    }
    const int nt = brief_nt(d);
    const int64_t FP = 32 * nt, npad = brief_npad_d(d, n), hidden = d.layers - 2 > 0 ? d.layers - 2 : 0;
    const bool small = use_small(d);
    WsLayout w;
    w.z = 0;
    w.dd = w.z + (small ? 0 : (hidden + (brief_use_wide(d) ? 1 : 0)) * FP * npad);      // k_wide stashes the last sine layer's phase too
    w.rec = w.dd + (small ? 0 : hidden * FP * npad);
    w.slabs = w.rec + rec_region_floats(d);
    w.table = w.slabs + hidden * (int64_t)(small ? small_grid(d, n) : wgrad_splits(d, n)) * (FP * FP + FP);
    w.table = (w.table + 3) / 4 * 4;
    w.img = w.table + (small ? kGroupTableFloats : 0);
    w.img = (w.img + 63) / 64 * 64;                                  // k_wide: two image-ordered ping-pong planes per workgroup (256-byte aligned)
    w.total = w.img + (brief_use_wide(d) ? (int64_t)kCUs * 2 * FP * 32 : 0);
    return w;
}

extern "C" {

int brief_version(void) { return BRIEF_VERSION; }
const char *brief_last_error(void) { return g_err; }

int64_t brief_param_count(const brief_siren_desc *d) { return check_desc(d) ? -1 : brief_canon_count(*d); }
int64_t brief_packed_count(const brief_siren_desc *d) { return check_desc(d) ? -1 : brief_pk_count(*d); }
int64_t brief_train_workspace_bytes(const brief_siren_desc *d, int64_t n)
{
    if (check_desc(d) || n < 1) return -1;
    return ws_layout(*d, n).total * (int64_t)sizeof(float);
}

int brief_siren_repack(const brief_siren_desc *d, const float *params, float *packed, void *stream)
{
    if (int rc = check_desc(d)) return rc;
    if (!params || !packed) return fail(BRIEF_ERR_INVALID, "null buffer");
    const int64_t total = brief_pk_count(*d);
    hipLaunchKernelGGL(k_repack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *d, params, packed);
    HIP_TRY(hipGetLastError());
    return 0;
}

}   // extern "C" (helpers below need C++ linkage)

static void fill_grid(GridArgs &g, const brief_grid_desc *grid)
{
    memset(&g, 0, sizeof(g));
    if (!grid) return;
    g.ndim = grid->ndim;
    g.lo = grid->lo; g.hi = grid->hi;
    double total = 1.0;
    for (int a = 0; a < 3; ++a) {
        g.dims[a] = a < grid->ndim ? grid->dims[a] : 1;
        g.step[a] = g.dims[a] > 1 ? (grid->hi - grid->lo) / (float)(g.dims[a] - 1) : 0.f;
        g.magic[a] = ~(uint64_t)0 / (uint64_t)g.dims[a] + 1;
        total *= (double)g.dims[a];
    }
    g.fast = total < 4294967296.0;
}

// One launch, optionally with a start / stop event pair bound to the DISPATCH itself (hipExtLaunchKernel: the timestamps are the
// kernel's own begin and end): the live timing of the dominant kernel (brief_profile_*) then costs no marker packets on the stream
// — with hipEventRecord around the launch every step carried two barrier packets, 1.8 % of the headline step.
template <typename KP, typename A>
static inline void launch_timed(KP kern, int grid, int block, size_t lds, hipStream_t st, const A &arg, hipEvent_t e0, hipEvent_t e1)
{
    if (e0 || e1) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), (uint32_t)lds, st, e0, e1, 0, arg);
    else hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, st, arg);
}

template <bool TRAIN>
static int launch_fused(const FusedArgs &fa, int grid, hipStream_t st, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr)
{
    const int nt = brief_nt(fa.d);
    if (fa.d.precision == BRIEF_PREC_BF16X3 && (TRAIN || g_x3_decode)) {
        // split precision: the 64-sample walk (k_fused_x3); BRIEF_X3_DECODE=0 evaluates such a net on the f32 forward kernel below
        const size_t lds = sizeof(float) * X3TLds::TOTAL;
        if (int rc = dev_attr_once((const void *)k_fused_x3<TRAIN>, (int)lds)) return rc;
        launch_timed(k_fused_x3<TRAIN>, grid, 256, lds, st, fa, e0, e1);
        HIP_TRY(hipGetLastError());
        return 0;
    }
#define BRIEF_CASE(NTV)                                                                                  \
    case NTV: {                                                                                          \
        const size_t lds = sizeof(float) * FusedLds<NTV, !TRAIN>::TOTAL;                                 \
        launch_timed(k_fused<NTV, TRAIN>, grid, 256, lds, st, fa, e0, e1);                               \
        break;                                                                                           \
    }
#if BRIEF_FUSED64
    if (nt == 8 && TRAIN) {
        // experiments on the 8-tile TRAIN step: 1 = 64-sample tiles (k_lean<2, 2, 8>: every weight fragment against two sample halves);
        // 2 = k_lean's one-state-array skeleton on 32-sample tiles, three workgroups per CU (k_lean<1, 2, 8>)
        constexpr int SHV = BRIEF_FUSED64 == 1 ? 2 : 1;
        const size_t lds = sizeof(float) * lean_lds(SHV, 2, 8).total;
        if (int rc = dev_attr_once((const void *)k_lean<SHV, 2, 8, TRAIN>, (int)lds)) return rc;
        launch_timed(k_lean<SHV, 2, 8, TRAIN>, grid, 256, lds, st, fa, e0, e1);
        HIP_TRY(hipGetLastError());
        return 0;
    }
#endif
    if (brief_use_wide(fa.d)) {
        // k_wide<MTW, TRAIN>: 33 .. 128 feature tiles, output tiles in ceil(nt / 4 MTW) passes of 4 waves x MTW slots, K-slabs staged from the image planes
        const int mtw = wide_mtw(nt);
        const size_t lds = sizeof(float) * wide_lds(mtw).total;
        if (!fa.Z) return fail(BRIEF_ERR_WORKSPACE, "widths above 1024 features evaluate through a scratch: call brief_siren_forward_ws with brief_forward_workspace_bytes() bytes");
#define BRIEF_WIDE_CASE(MTWV)                                                                            \
    case MTWV:                                                                                           \
        if (int rc = dev_attr_once((const void *)k_wide<MTWV, TRAIN>, (int)lds)) return rc;              \
        launch_timed(k_wide<MTWV, TRAIN>, grid, 256, lds, st, fa, e0, e1);                               \
        break;
        switch (mtw) {
            BRIEF_WIDE_CASE(3) BRIEF_WIDE_CASE(4) BRIEF_WIDE_CASE(5) BRIEF_WIDE_CASE(6) BRIEF_WIDE_CASE(7) BRIEF_WIDE_CASE(8)
        default: return fail(BRIEF_ERR_INVALID, "unsupported width");
        }
#undef BRIEF_WIDE_CASE
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (brief_use_lean(fa.d, TRAIN)) {
        // k_lean<1, MTW, 0>: a run-time number of feature tiles, MTW = ceil(nt / 4) of them per wave (brief_layout.h: brief_use_lean says
        // which widths; inference of exactly 12 or 16 tiles stays on k_fused<12 / 16, false>, whose unrolled chains decode 5 % faster:
        // 4x384 0.89 against 0.84 of the fp32 peak, 4x512 0.92 against 0.87 — tools/decode_widths.py; the layouts are the same)
        const int mtw = (nt + 3) / 4;
        const int rm = BRIEF_LEAN_RM3 ? (nt & 3) : ((nt & 3) == 3 ? 0 : (nt & 3));      // left-over tiles shared along K by the four waves (brief_lean.inc)
        const size_t lds = sizeof(float) * lean_lds(1, mtw, nt).total;
#define BRIEF_WIDE_RM(MTWV, RMV)                                                                         \
    {                                                                                                    \
        if (int rc = dev_attr_once((const void *)k_lean<1, MTWV, 0, TRAIN, RMV>, (int)(sizeof(float) * lean_lds(1, MTWV, 4 * MTWV).total))) return rc; \
        launch_timed(k_lean<1, MTWV, 0, TRAIN, RMV>, grid, 256, lds, st, fa, e0, e1);                    \
    }
#define BRIEF_WIDE(MTWV)                                                                                 \
    case MTWV:                                                                                           \
        if (rm == 1) BRIEF_WIDE_RM(MTWV, 1) else if (rm == 2) BRIEF_WIDE_RM(MTWV, 2) else if (rm == 3) BRIEF_WIDE_RM(MTWV, 3) else BRIEF_WIDE_RM(MTWV, 0)   \
        break;
        switch (mtw) {
            case 1: BRIEF_WIDE_RM(1, 3) break;   // 3 tiles (TRAIN): all three shared along K by the four waves
            BRIEF_WIDE(2)                    // 5 .. 7 tiles
            BRIEF_WIDE(3) BRIEF_WIDE(4)      // 257 .. 512 features
            BRIEF_WIDE(5) BRIEF_WIDE(6) BRIEF_WIDE(7) BRIEF_WIDE(8)
        default: return fail(BRIEF_ERR_INVALID, "unsupported width");
        }
#undef BRIEF_WIDE
#undef BRIEF_WIDE_RM
        HIP_TRY(hipGetLastError());
        return 0;
    }
    switch (nt) {
        BRIEF_CASE(1) BRIEF_CASE(2) BRIEF_CASE(3) BRIEF_CASE(4) BRIEF_CASE(8)
    case 12:
    case 16:
        if constexpr (!TRAIN) {      // inference only (their TRAIN steps run on k_lean)
            const size_t lds12 = sizeof(float) * FusedLds<12, true>::TOTAL, lds16 = sizeof(float) * FusedLds<16, true>::TOTAL;
            if (nt == 12) launch_timed(k_fused<12, false>, grid, 256, lds12, st, fa, e0, e1);
            else launch_timed(k_fused<16, false>, grid, 256, lds16, st, fa, e0, e1);
            break;
        }
        return fail(BRIEF_ERR_INVALID, "unsupported width");
    default: return fail(BRIEF_ERR_INVALID, "unsupported width");
    }
#undef BRIEF_CASE
    HIP_TRY(hipGetLastError());
    return 0;
}

template <bool TRAIN>
static int launch_k16(const FusedArgs &fa, int grid, hipStream_t st, int ns)
{
    const int nt = brief_nt(fa.d);
    bool launched = false;
#define BRIEF_CASE(NTV, COV, NSV)                                                                        \
    if (!launched && nt == NTV && (fa.d.cout == 1) == (COV == 1) && ns == NSV) {                         \
        const size_t lds = sizeof(float) * Cfg16<NTV, NSV>::TOTAL;                                       \
        if (int rc = dev_attr_once((const void *)k16<NTV, TRAIN, COV, NSV>, (int)lds)) return rc;          \
        hipLaunchKernelGGL((k16<NTV, TRAIN, COV, NSV>), dim3(grid), dim3(512), lds, st, fa);             \
        launched = true;                                                                                 \
    }
    if (nt != 8 && nt != 16) return fail(BRIEF_ERR_INVALID, "unsupported width");
    BRIEF_CASE(8, 1, 4) BRIEF_CASE(8, 1, 2) BRIEF_CASE(8, 1, 1) BRIEF_CASE(8, 4, 4)
    BRIEF_CASE(16, 1, 4) BRIEF_CASE(16, 1, 2) BRIEF_CASE(16, 1, 1) BRIEF_CASE(16, 4, 4)
#undef BRIEF_CASE
    if (!launched) return fail(BRIEF_ERR_INVALID, "no bf16 kernel for this configuration");
    HIP_TRY(hipGetLastError());
    return 0;
}

template <bool TRAIN>
static int launch_k16_split(FusedArgs &fa, hipStream_t st)
{
    const int64_t np = npad16(fa.n);
    const Split16 sp = split16(fa.d, fa.n);
    if (sp.g_body > 0) {
        fa.n_begin = 0; fa.n_end = sp.g_tail > 0 ? sp.n_body : np; fa.rec_base = 0;
        if (int rc = launch_k16<TRAIN>(fa, sp.g_body, st, 4)) return rc;
    }
    if (sp.g_tail > 0) {
        fa.n_begin = sp.n_body; fa.n_end = np; fa.rec_base = sp.g_body;
        if (int rc = launch_k16<TRAIN>(fa, sp.g_tail, st, sp.ns_tail)) return rc;
    }
    return 0;
}

// the live timing is on, has slots left, and was enabled on the device this call runs on
static bool prof_live()
{
    return g_prof_on && g_prof_n < kProfSlots && g_prof_dev == current_device() && dev_state() && dev_state()->prof_init;
}

static int check_batch(const brief_siren_desc *d, const brief_grid_desc *grid, const brief_batch_desc *b, bool train)
{
    if (!b || b->n < 1) return fail(BRIEF_ERR_INVALID, "empty batch");
    if (!b->coords) {
        if (!grid || grid->ndim != d->cin) return fail(BRIEF_ERR_INVALID, "grid.ndim must equal coords_channel when coords is NULL");
        for (int a = 0; a < grid->ndim; ++a)
            if (grid->dims[a] < 1) return fail(BRIEF_ERR_INVALID, "bad grid dims");
    }
    if (train && !b->targets) return fail(BRIEF_ERR_INVALID, "targets required");
    return 0;
}

extern "C" {

// scratch of an inference launch: two ping-pong activation planes ([FP rows][32 samples]) per workgroup for k_wide, nothing otherwise
int64_t brief_forward_workspace_bytes(const brief_siren_desc *d, int64_t n)
{
    if (check_desc(d) || n < 1) return -1;
    if (!brief_use_wide(*d)) return 0;
    return (int64_t)fused_grid(*d, n, false) * 2 * 32 * brief_nt(*d) * 32 * (int64_t)sizeof(float);
}

int brief_siren_forward(const brief_siren_desc *d, const float *packed, const brief_grid_desc *grid,
                        const brief_batch_desc *batch, void *out, int out_kind,
                        float scale_min, float scale_max, double vmin, double vmax, void *stream)
{
    return brief_siren_forward_ws(d, packed, grid, batch, out, out_kind, scale_min, scale_max, vmin, vmax, nullptr, 0, stream);
}

int brief_siren_forward_ws(const brief_siren_desc *d, const float *packed, const brief_grid_desc *grid,
                           const brief_batch_desc *batch, void *out, int out_kind,
                           float scale_min, float scale_max, double vmin, double vmax, void *workspace, int64_t workspace_bytes, void *stream)
{
    if (int rc = check_desc(d)) return rc;
    if (int rc = check_batch(d, grid, batch, false)) return rc;
    if (brief_use_wide(*d)) {
        if (!workspace || workspace_bytes < brief_forward_workspace_bytes(d, batch->n))
            return fail(BRIEF_ERR_WORKSPACE, "widths above 1024 features evaluate through a scratch of brief_forward_workspace_bytes() bytes");
    }
    if (!packed || !out) return fail(BRIEF_ERR_INVALID, "null buffer");
    if (out_kind < BRIEF_OUT_F32 || out_kind > BRIEF_OUT_U16) return fail(BRIEF_ERR_INVALID, "bad out_kind");
    FusedArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.d = *d; fa.pk = packed;
    fa.coords = batch->coords; fa.idx = batch->idx; fa.offset = batch->offset; fa.n = batch->n;
    fill_grid(fa.grid, grid);
    fa.npad = brief_npad(brief_nt(*d), batch->n);
    fa.out = out; fa.out_kind = out_kind;
    fa.scale_min = scale_min;
    fa.den = (float)((double)scale_max - (double)scale_min);
    fa.span = (float)(vmax - vmin);
    fa.vmin = (float)vmin;
    fa.stagger_cus = kCUs; fa.stagger = 0;
    if (brief_use_wide(*d)) fa.Z = (float *)workspace;      // k_wide<.., false>: the ping-pong planes
    if (d->precision == BRIEF_PREC_BF16) return launch_k16_split<false>(fa, (hipStream_t)stream);
    const FusedPlan fp = fused_plan(*d, batch->n, false);
    fa.pers_wgs = fp.pers_wgs; fa.pers_tiles = fp.pers_tiles;
    return launch_fused<false>(fa, fp.grid, (hipStream_t)stream);
}

struct UpdatePayload { OptimScalars opt; float *params, *s1, *s2, *pk; };

}   // extern "C"

// what one narrow-net step launches, as data: brief_multi_fit collects these for a group of jobs and launches them together
struct SmallStepPlan { FusedArgs fa; ReduceArgs ra; int grid1, nb_hidden, nb_reduce, nt, hb; };

// the weight-gradient GEMM of the hidden layers: `blocks` = K splits of this launch x hidden layers x quadrants
static int launch_wgrad(const brief_siren_desc &d, const WgradArgs &wa, int blocks, hipStream_t st)
{
    const int nt = brief_nt(d);
    if (d.precision == BRIEF_PREC_BF16X3) {
        for (int rep = 0; rep < g_wgrad_repeat; ++rep)      // BRIEF_WGRAD_REPEAT (diagnostics): the launch is idempotent
            hipLaunchKernelGGL(k_wgrad_x3, dim3(blocks), dim3(512), sizeof(float) * wgrad_lds_floats(8), st, wa);
    } else if (nt > 8) {
        // run-time width: ceil(nt / 8) quadrants per side of QT = ceil(nt / quadrants) tiles (the last row / column may be short)
        const int qt = (nt + wgrad_nq(nt) - 1) / wgrad_nq(nt);
        switch (qt) {
        case 5: hipLaunchKernelGGL((k_wgrad<0, 5>), dim3(blocks), dim3(512), sizeof(float) * wgrad_lds_floats(5), st, wa); break;
        case 6: hipLaunchKernelGGL((k_wgrad<0, 6>), dim3(blocks), dim3(512), sizeof(float) * wgrad_lds_floats(6), st, wa); break;
        case 7: hipLaunchKernelGGL((k_wgrad<0, 7>), dim3(blocks), dim3(512), sizeof(float) * wgrad_lds_floats(7), st, wa); break;
        case 8: hipLaunchKernelGGL((k_wgrad<0, 8>), dim3(blocks), dim3(512), sizeof(float) * wgrad_lds_floats(8), st, wa); break;
        default: return fail(BRIEF_ERR_INVALID, "unsupported width");
        }
    } else {
#define BRIEF_CASE(NTV)                                                                                    \
    case NTV:                                                                                              \
        hipLaunchKernelGGL((k_wgrad<NTV>), dim3(blocks), dim3(512), sizeof(float) * wgrad_lds_floats(NTV), st, wa); \
        break;
        switch (nt) {
            BRIEF_CASE(1) BRIEF_CASE(2) BRIEF_CASE(3) BRIEF_CASE(4)
            BRIEF_CASE(5) BRIEF_CASE(6) BRIEF_CASE(7) BRIEF_CASE(8)
        }
#undef BRIEF_CASE
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

static int train_step_impl(const brief_siren_desc *d, const float *packed, const brief_grid_desc *grid,
                           const brief_batch_desc *batch, int loss_kind, float thr, float beta,
                           float *grads, float *loss_out, float *yhat_out,
                           void *workspace, int64_t workspace_bytes, void *stream, const UpdatePayload *upd, SmallStepPlan *plan_only = nullptr)
{
    std::lock_guard<std::recursive_mutex> state_lock(g_state_mu);      // (the side-stream plans below: see g_state_mu)
    if (int rc = check_desc(d)) return rc;
    if (int rc = check_batch(d, grid, batch, true)) return rc;
    if (!packed || !grads || !loss_out || !workspace) return fail(BRIEF_ERR_INVALID, "null buffer");
    if (loss_kind < BRIEF_LOSS_L2 || loss_kind > BRIEF_LOSS_EXTERNAL) return fail(BRIEF_ERR_INVALID, "bad loss_kind");
    if (d->precision == BRIEF_PREC_BF16 ? (int64_t)32 * brief_nt(*d) * npad16(batch->n) * 2 >= ((int64_t)1 << 31)
                                       : (int64_t)32 * brief_nt(*d) * brief_npad_d(*d, batch->n) * 4 >= ((int64_t)1 << 31))
        return fail(BRIEF_ERR_INVALID, "batch too large for one train step (padded width x samples x 4 bytes must stay below 2 GiB): split it");
    const WsLayout wl = ws_layout(*d, batch->n);
    if (workspace_bytes < wl.total * (int64_t)sizeof(float)) return fail(BRIEF_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float *ws = (float *)workspace;
    const int nt = brief_nt(*d);
    if (d->precision == BRIEF_PREC_BF16) {
        const Ws16 w16 = ws16_layout(*d, batch->n);
        const int64_t np = npad16(batch->n), FP = 32 * nt;
        const int hidden = d->layers - 2;
        const Split16 sp16 = split16(*d, batch->n);
        const int g16 = sp16.g_body + sp16.g_tail, nsp = wgrad16_splits(*d, batch->n);
        const float inv16 = (float)(1.0 / ((double)batch->n * d->cout));
        FusedArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.d = *d; fa.pk = packed;
        fa.coords = batch->coords; fa.targets = batch->targets; fa.weights = batch->weights;
        fa.idx = batch->idx; fa.offset = batch->offset; fa.n = batch->n;
        if (!batch->idx && batch->rng_pop > 0) { fa.rng_pop = (uint64_t)batch->rng_pop; fa.rng_seed = batch->rng_seed; fa.rng_step = batch->rng_step; }
        fill_grid(fa.grid, grid);
        fa.loss_kind = loss_kind; fa.thr = thr; fa.beta = beta; fa.inv_count = inv16;
        fa.npad = np; fa.rec = ws + w16.rec; fa.yhat_out = yhat_out; fa.diag = g_diag;
        fa.stagger_cus = kCUs; fa.stagger = g_stagger;
        fa.S16[0] = ws + w16.h; fa.S16[1] = nullptr; fa.S16[2] = ws + w16.dd; fa.S16[3] = ws + w16.x; fa.S16[4] = ws + w16.g;
        const bool prof16 = prof_live();
        hipEvent_t *pev16 = prof16 ? dev_state()->prof_ev : nullptr;
        const Split16Plan ov = split16_plan(*d, batch->n);      // overlap plan (512-wide nets with a tail launch): see split16_plan
        hipStream_t side = st;
        if (ov.on) {
            DevState *ds = dev_state();
            if (!ds) return fail(BRIEF_ERR_INVALID, "out of host memory");
            if (!ds->aux_init) {
                int least = 0, greatest = 0;
                HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
                HIP_TRY(hipStreamCreateWithPriority(&ds->aux, hipStreamNonBlocking, least));      // the main launch's workgroups go first
                HIP_TRY(hipEventCreateWithFlags(&ds->aux_fork, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&ds->aux_join, hipEventDisableTiming));
                ds->aux_init = true;
            }
            side = ds->aux;
        }
        if (prof16) HIP_TRY(hipEventRecord(pev16[2 * g_prof_n], st));
        if (!ov.on) {
            if (int rc = launch_k16_split<true>(fa, st)) return rc;
        } else {
            fa.n_begin = 0; fa.n_end = sp16.n_body; fa.rec_base = 0;
            if (int rc = launch_k16<true>(fa, sp16.g_body, st, 4)) return rc;
            HIP_TRY(hipEventRecord(dev_state()->aux_fork, st));
            HIP_TRY(hipStreamWaitEvent(side, dev_state()->aux_fork, 0));
            fa.n_begin = sp16.n_body; fa.n_end = np; fa.rec_base = sp16.g_body;
            if (int rc = launch_k16<true>(fa, sp16.g_tail, side, sp16.ns_tail)) return rc;
        }
        if (prof16) { HIP_TRY(hipEventRecord(pev16[2 * g_prof_n + 1], side)); ++g_prof_n; }      // (overlap plan: from the body's start to the tail's end)
        Wgrad16Args wa;
        memset(&wa, 0, sizeof(wa));
        const int nsp_s = wgrad16_skinny_splits(*d, batch->n);
        wa.d = *d; wa.npad = np; wa.nsplit = nsp; wa.nsplit_s = nsp_s; wa.bias_jobs = 0; wa.slabs = ws + w16.slabs;   // (bias_jobs: k_wgrad16_big computes db_l itself since round 2)
        wa.H = (const __bf16 *)(ws + w16.h); wa.D = (const __bf16 *)(ws + w16.dd);
        wa.X = (const __bf16 *)(ws + w16.x); wa.G = (const __bf16 *)(ws + w16.g);
        wa.kb_lo = 0; wa.kb_hi = np / 64; wa.split_lo = 0; wa.nsplit_here = nsp;
        const int nb = nt / 4;
        if (hidden > 0 && nt == 16) {
            const int lds_big = (int)(sizeof(float) * 4 * W16B_PANEL);
            if (int rc = dev_attr_once((const void *)k_wgrad16_big, lds_big)) return rc;
            if (ov.on) {
                // the body's blocks in ov.na splits on the caller's stream, the tail's blocks as split number ov.na on the side stream
                wa.kb_lo = 0; wa.kb_hi = ov.kb_body; wa.split_lo = 0; wa.nsplit_here = ov.na;
                hipLaunchKernelGGL(k_wgrad16_big, dim3((hidden * ov.na + 7) / 8 * 8 * 4), dim3(512), lds_big, st, wa);
                wa.kb_lo = ov.kb_body; wa.kb_hi = np / 64; wa.split_lo = ov.na; wa.nsplit_here = 1;
                hipLaunchKernelGGL(k_wgrad16_big, dim3((hidden + 7) / 8 * 8 * 4), dim3(512), lds_big, side, wa);
            } else {
                const int groups8 = (hidden * nsp + 7) / 8 * 8;        // (layer, split) groups padded to whole XCD rounds (see the kernel)
                hipLaunchKernelGGL(k_wgrad16_big, dim3(groups8 * 4), dim3(512), lds_big, st, wa);
            }
        } else if (hidden > 0) {
            hipLaunchKernelGGL(k_wgrad16<false>, dim3(hidden * nb * nb * nsp), dim3(256), sizeof(float) * 4 * W16_PANEL, st, wa);
        }
        // hidden-layer bias gradients (B = ones), first layer, head
        hipLaunchKernelGGL(k_wgrad16<true>, dim3((wa.bias_jobs ? hidden * nb * nsp : 0) + 2 * nb * nsp_s), dim3(256), sizeof(float) * 4 * W16_PANEL, side, wa);
        HIP_TRY(hipGetLastError());
        if (ov.on) {
            HIP_TRY(hipEventRecord(dev_state()->aux_join, side));
            HIP_TRY(hipStreamWaitEvent(st, dev_state()->aux_join, 0));
        }
        const int64_t l0c = (int64_t)d->features * d->cin + d->features;
        const int64_t hcnt = brief_canon_head_off(*d) - l0c;
        if (hidden > 0) {
            ReduceArgs ra;
            memset(&ra, 0, sizeof(ra));
            ra.d = *d; ra.slabs = ws + w16.slabs; ra.nsplit = nsp; ra.sgroups = 1;
            ra.grads = grads; ra.loss_out = loss_out; ra.inv_count = inv16;
            if (upd) { ra.update = 1; ra.opt = upd->opt; ra.params = upd->params; ra.s1 = upd->s1; ra.s2 = upd->s2; ra.pk = upd->pk; }
            // one thread per parameter: k_wgrad16 leaves only a handful of slabs per layer (8x512: 39.6 us against 63.2 us
            // with 4 threads per parameter and 106 us with 8; 4x256: 8.5 / 10.8 us)
            ra.sgroups = 1;
            const int nbh = (int)((hcnt + 255) / 256);
            hipLaunchKernelGGL(k_reduce, dim3(nbh), dim3(256), 0, st, ra, nbh);      // hidden-layer part only
            HIP_TRY(hipGetLastError());
        }
        Reduce16Args r16;
        memset(&r16, 0, sizeof(r16));
        r16.d = *d; r16.slabs = ws + w16.slabs + (int64_t)hidden * nsp * (FP * FP + FP); r16.nsplit = nsp_s;
        r16.rec = fa.rec; r16.nwg = g16; r16.grads = grads; r16.loss_out = loss_out; r16.inv_count = inv16;
        if (upd) { r16.update = 1; r16.opt = upd->opt; r16.params = upd->params; r16.s1 = upd->s1; r16.s2 = upd->s2; r16.pk = upd->pk; }
        const int64_t items = l0c + (int64_t)d->cout * d->features + d->cout + 1;
        hipLaunchKernelGGL(k_reduce16, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, r16);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const bool small = use_small(*d);
    const FusedPlan fp = fused_plan(*d, batch->n, true);
    TailPlan tp;      // fp32 nets of 8 tiles and more: the last round's tiles as a launch of their own (fused_tail_plan)
    memset(&tp, 0, sizeof(tp));
    if (!small) tp = fused_tail_plan(*d, batch->n);
    const int grid1 = small ? small_grid(*d, batch->n) : (tp.on ? tp.grid_body + tp.grid_tail : fp.grid);      // (= workgroup records)
    const int nsplit = small ? (d->layers > 2 ? grid1 : 0) : wgrad_splits(*d, batch->n);
    const float inv_count = (float)(1.0 / ((double)batch->n * d->cout));

    FusedArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.d = *d; fa.pk = packed;
    fa.coords = batch->coords; fa.targets = batch->targets; fa.weights = batch->weights;
    fa.idx = batch->idx; fa.offset = batch->offset; fa.n = batch->n;
    if (!batch->idx && batch->rng_pop > 0) { fa.rng_pop = (uint64_t)batch->rng_pop; fa.rng_seed = batch->rng_seed; fa.rng_step = batch->rng_step; }
    fill_grid(fa.grid, grid);
    fa.loss_kind = loss_kind; fa.thr = thr; fa.beta = beta; fa.inv_count = inv_count;
    fa.Z = ws + wl.z; fa.D = ws + wl.dd; fa.npad = brief_npad_d(*d, batch->n);
    fa.rec = ws + wl.rec; fa.slabs = brief_use_wide(*d) ? ws + wl.img : ws + wl.slabs; fa.yhat_out = yhat_out;      // (k_small: its slabs; k_wide: its image planes)
    fa.stagger_cus = kCUs; fa.stagger = g_stagger; fa.diag = g_diag;
    fa.pers_wgs = fp.pers_wgs; fa.pers_tiles = fp.pers_tiles;
    ReduceArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.d = *d; ra.rec = fa.rec; ra.nrec_wg = grid1; ra.slabs = ws + wl.slabs; ra.nsplit = nsplit;
    // threads per hidden parameter: k_small leaves one slab per workgroup (<= 2 x CUs of them): 16 slabs per thread, i.e.
    // 32 threads per parameter for 512 slabs — every thread's loads are then ONE round trip to HBM instead of eight (measured,
    // tools/ab_reduce.sh: 4x22 on 64^3 0.084 -> 0.072 ms per step, 2x64 0.111 -> 0.103; 64 threads per parameter: 0.076).
    // k_wgrad <= 85 slabs: 4 threads x ~21.  The group sums are added in group order (bit-reproducible).
    int sg_small = 8;
    while (sg_small < 32 && sg_small * 16 < nsplit) sg_small *= 2;
    if (g_reduce_sg_small > 0) sg_small = g_reduce_sg_small;
    ra.sgroups = small ? sg_small : g_reduce_sg_big;
    const int rthreads = 256;
    ra.grads = grads; ra.loss_out = loss_out; ra.inv_count = inv_count;
    if (upd) { ra.update = 1; ra.opt = upd->opt; ra.params = upd->params; ra.s1 = upd->s1; ra.s2 = upd->s2; ra.pk = upd->pk; }
    const int64_t l0_count = (int64_t)d->features * d->cin + d->features;
    const int64_t hcount = brief_canon_head_off(*d) - l0_count;
    const int64_t skinny = d->features + (int64_t)d->cout * ((d->features + 3) / 4) + 2;   // groups of four record slots (reduce_body): first-layer rows, head quads, head biases, the loss
    // fp32 nets behind k_wgrad: 4 / 2 consecutive parameters per thread where the width is a multiple (reduce_hidden_vec)
    ra.vec = (!small && g_reduce_vec && d->precision == BRIEF_PREC_F32 && rthreads * 4 <= 1024) ? (d->features % 4 == 0 ? 4 : (d->features % 2 == 0 ? 2 : 0)) : 0;
    {   // (caller-owned buffers: fall back to narrower accesses when one is not aligned for them)
        const uintptr_t bits = (uintptr_t)grads | (uintptr_t)ra.params | (uintptr_t)ra.s1 | (uintptr_t)ra.s2 | (uintptr_t)ra.pk | (uintptr_t)ra.slabs;
        while (ra.vec > 1 && bits % (sizeof(float) * ra.vec) != 0) ra.vec /= 2;
        if (ra.vec == 1) ra.vec = 0;
    }
    const int ppb = rthreads / ra.sgroups * (ra.vec ? ra.vec : 1);             // hidden parameters per k_reduce block
    const int nb_hidden = (int)((hcount + ppb - 1) / ppb);
    const int wpb = rthreads / 64;                     // skinny items (one wave each) per block
    const int nb_skinny = (int)((skinny + wpb - 1) / wpb);
    if (plan_only) {
        if (!small) return fail(BRIEF_ERR_INVALID, "internal: step plans exist for narrow nets only");
        plan_only->fa = fa; plan_only->ra = ra; plan_only->grid1 = grid1; plan_only->nb_hidden = nb_hidden;
        plan_only->nb_reduce = nb_hidden + nb_skinny; plan_only->nt = nt; plan_only->hb = small_hb(*d);
        return 0;
    }
    const bool prof = prof_live();
    hipEvent_t pe0 = prof ? dev_state()->prof_ev[2 * g_prof_n] : nullptr, pe1 = prof ? dev_state()->prof_ev[2 * g_prof_n + 1] : nullptr;
    if (small) {
        const int hb = small_hb(*d);
#define BRIEF_CASE(NTV, HBV)                                                                                        \
    if (nt == NTV && hb == HBV)                                                                                     \
        launch_timed(k_small<NTV, HBV>, grid1, 256, sizeof(float) * SmallLds<NTV>::TOTAL, st, fa, pe0, pe1);
        BRIEF_CASE(1, 1) BRIEF_CASE(1, 3) BRIEF_CASE(1, 5) BRIEF_CASE(1, 7)
        BRIEF_CASE(2, 1) BRIEF_CASE(2, 3) BRIEF_CASE(2, 5) BRIEF_CASE(2, 7)
#undef BRIEF_CASE
        HIP_TRY(hipGetLastError());
    } else if (!tp.on) {
        if (int rc = launch_fused<true>(fa, grid1, st, pe0, pe1)) return rc;
    }
    hipStream_t side = st;
    if (tp.on) {
        DevState *ds = dev_state();
        if (!ds) return fail(BRIEF_ERR_INVALID, "out of host memory");
        if (!ds->aux2_init) {
            HIP_TRY(hipStreamCreateWithFlags(&ds->aux2, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&ds->aux2_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&ds->aux2_join, hipEventDisableTiming));
            ds->aux2_init = true;
        }
        side = ds->aux2;
        fa.n_begin = 0; fa.n_end = tp.tiles_body * 32; fa.rec_base = 0;      // (k_lean, k_wide)
        fa.pers_wgs = tp.grid_body; fa.pers_tiles = tp.tiles_body;           // (k_fused<8>: persistent workgroups over the body's tiles)
        if (int rc = launch_fused<true>(fa, tp.grid_body, st, pe0, nullptr)) return rc;
        HIP_TRY(hipEventRecord(ds->aux2_fork, st));
        HIP_TRY(hipStreamWaitEvent(side, ds->aux2_fork, 0));
        // the tail stays on the caller's stream, right behind the body (its workgroups are placed first); the body's k_wgrad arrives through the event a moment later
        fa.n_begin = tp.tiles_body * 32; fa.n_end = 0; fa.rec_base = tp.grid_body;
        fa.pers_wgs = 0;                                                     // (k_fused<8>: one tile per workgroup, from pers_tiles on)
        if (int rc = launch_fused<true>(fa, tp.grid_tail, st, nullptr, pe1)) return rc;      // (the profile pair: body start .. tail end)
    }
    if (prof) ++g_prof_n;

    if (!small && nsplit > 0) {
        WgradArgs wa;
        memset(&wa, 0, sizeof(wa));
        wa.d = *d; wa.Z = fa.Z; wa.D = fa.D; wa.npad = fa.npad; wa.nsplit = nsplit; wa.slabs = ws + wl.slabs;
        wa.stamps = ws + wl.rec + rec_region_floats(*d) - 256 * 8 * 8;   // tail of the record region (diagnostics)
        const int per_split = (d->layers - 2) * wgrad_nq(nt) * wgrad_nq(nt);
        if (tp.mode == 2) {
            // uneven splits: the normal ones on the side stream behind the body, the short ones (which end with the tail's chunks) on the caller's behind the tail
            wa.uneven = 1; wa.n_norm = tp.n_norm; wa.kfix_n = tp.kfix_n; wa.kfix_s = tp.kfix_s;
            wa.kb_lo = 0; wa.kb_hi = fa.npad / 32;
            wa.split_lo = 0; wa.nsplit_here = tp.n_norm;
            if (int rc = launch_wgrad(*d, wa, wa.nsplit_here * per_split, side)) return rc;
            wa.split_lo = tp.n_norm; wa.nsplit_here = nsplit - tp.n_norm;
            if (int rc = launch_wgrad(*d, wa, wa.nsplit_here * per_split, st)) return rc;
        } else {
            // K range and slabs of a launch: every chunk into nsplit slabs, or (extra-split tail plan) the body's chunks into nsplit - 1 and the tail's into the last one
            const int nsA = tp.on ? nsplit - 1 : nsplit;
            wa.kb_lo = 0; wa.kb_hi = tp.on ? tp.tiles_body : fa.npad / 32; wa.split_lo = 0; wa.nsplit_here = nsA;
            for (int part = 0; part < (tp.on ? 2 : 1); ++part) {
                if (part) { wa.kb_lo = tp.tiles_body; wa.kb_hi = fa.npad / 32; wa.split_lo = nsA; wa.nsplit_here = 1; }
                // (the body's chunks on the side stream, the tail's — behind the tail launch — on the caller's)
                if (int rc = launch_wgrad(*d, wa, wa.nsplit_here * per_split, part || !tp.on ? st : side)) return rc;
            }
        }
    }
    if (tp.on) {
        HIP_TRY(hipEventRecord(dev_state()->aux2_join, side));
        HIP_TRY(hipStreamWaitEvent(st, dev_state()->aux2_join, 0));
    }
    if (g_reduce_part == 1) hipLaunchKernelGGL(k_reduce, dim3(nb_hidden), dim3(rthreads), 0, st, ra, nb_hidden);          // diagnostics: the hidden-parameter blocks alone
    else if (g_reduce_part == 2) hipLaunchKernelGGL(k_reduce, dim3(nb_skinny), dim3(rthreads), 0, st, ra, 0);         // diagnostics: the skinny blocks alone
    else hipLaunchKernelGGL(k_reduce, dim3(nb_hidden + nb_skinny), dim3(rthreads), 0, st, ra, nb_hidden);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" {

int brief_siren_train_step(const brief_siren_desc *d, const float *packed, const brief_grid_desc *grid,
                           const brief_batch_desc *batch, int loss_kind, float thr, float beta,
                           float *grads, float *loss_out, float *yhat_out,
                           void *workspace, int64_t workspace_bytes, void *stream)
{
    return train_step_impl(d, packed, grid, batch, loss_kind, thr, beta, grads, loss_out, yhat_out, workspace, workspace_bytes,
                           stream, nullptr);
}

int brief_siren_fit_step(const brief_siren_desc *d, float *params, float *packed, const brief_grid_desc *grid,
                         const brief_batch_desc *batch, int loss_kind, float thr, float beta,
                         int optim_kind, float *state1, float *state2, double lr, double beta1, double beta2, double eps, int64_t t,
                         float *grads, float *loss_out, void *workspace, int64_t workspace_bytes, void *stream)
{
    if (optim_kind < BRIEF_OPT_ADAMAX || optim_kind > BRIEF_OPT_SGD) return fail(BRIEF_ERR_INVALID, "bad optimizer kind");
    if (!params || t < 1) return fail(BRIEF_ERR_INVALID, "bad optimizer arguments");
    if (optim_kind != BRIEF_OPT_SGD && (!state1 || !state2)) return fail(BRIEF_ERR_INVALID, "optimizer state required");
    UpdatePayload up;
    up.opt = optim_scalars(optim_kind, lr, beta1, beta2, eps, t);
    up.params = params; up.s1 = state1; up.s2 = state2; up.pk = packed;
    return train_step_impl(d, packed, grid, batch, loss_kind, thr, beta, grads, loss_out, nullptr, workspace, workspace_bytes, stream, &up);
}

static int fit_job_check(const brief_fit_job *j)
{
    if (!j) return fail(BRIEF_ERR_INVALID, "null job");
    if (j->batch.idx && j->idx_stride <= 0) return fail(BRIEF_ERR_INVALID, "brief_siren_fit needs idx_stride > 0 with batch.idx (one index set per step); a single set goes to brief_siren_fit_step");
    if (j->batch.idx && j->idx_stride < j->batch.n) return fail(BRIEF_ERR_INVALID, "idx_stride is smaller than the batch");
    if (!j->params || !j->packed || !j->grads || !j->loss_out || !j->workspace) return fail(BRIEF_ERR_INVALID, "null buffer");
    if (j->t0 < 0) return fail(BRIEF_ERR_INVALID, "bad step count");
    if (j->n_milestones < 0 || (j->n_milestones > 0 && !j->milestones)) return fail(BRIEF_ERR_INVALID, "bad lr milestones");
    return 0;
}

// the learning rate in force for optimizer step t (1-based; entry k of this call): *lr is the running MultiStepLR value
static void fit_job_lr(const brief_fit_job *j, int64_t t, int64_t k, double *lr)
{
    // scheduler.step() calls made so far = t - 1: apply the milestones that were hit by the last one
    if (j->lr_table) *lr = j->lr_table[k];
    else if (t - 1 > j->t0) {
        int hits = 0;
        for (int m = 0; m < j->n_milestones; ++m) hits += (j->milestones[m] == t - 1);
        if (hits) *lr = *lr * pow(j->gamma, (double)hits);
    }
}

// one optimizer step (number t, 1-based) of a job on stream st
static int fit_job_step(const brief_fit_job *j, int64_t t, int64_t k, double *lr, hipStream_t st)
{
    fit_job_lr(j, t, k, lr);
    brief_batch_desc b = j->batch;
    if (b.idx) b.idx = b.idx + k * j->idx_stride;          // device-resident index stream: this step's set
    else if (b.rng_pop > 0) b.rng_step = (uint64_t)t;
    return brief_siren_fit_step(&j->desc, j->params, j->packed, &j->grid, &b, j->loss_kind, j->thr, j->beta, j->optim_kind,
                                j->state1, j->state2, *lr, j->beta1_table ? j->beta1_table[k] : j->beta1, j->beta2, j->eps, t, j->grads,
                                j->loss_log ? j->loss_log + k : j->loss_out, j->workspace, j->workspace_bytes, (void *)st);
}

int brief_siren_fit(const brief_fit_job *job, int64_t steps, void *stream)
{
    if (int rc = fit_job_check(job)) return rc;
    if (steps < 0) return fail(BRIEF_ERR_INVALID, "bad step count");
    double lr = job->lr;
    for (int64_t k = 0; k < steps; ++k)
        if (int rc = fit_job_step(job, job->t0 + 1 + k, k, &lr, (hipStream_t)stream)) return rc;
    if (job->loss_log && steps > 0)
        HIP_TRY(hipMemcpyAsync(job->loss_out, job->loss_log + steps - 1, sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

// ---- co-trained narrow nets: one k_small_group + one k_reduce_group launch per step for a whole group of jobs -------------------
}   // extern "C" (templates need C++ linkage)
template <typename T>
__global__ void k_put(T *dst, const T v)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = v;
}
extern "C" {

struct FitGroup { int nt, hb, njobs, jobs[BRIEF_GROUP_MAX]; int small_grid_total, reduce_grid_total; FusedArgs *fa_table; ReduceArgs *ra_table;
                  SmallGroupArgs sg; ReduceGroupArgs rg; };

// the job's step as data (what brief_siren_fit_step would launch), validated the same way
static int fit_job_plan(const brief_fit_job *j, SmallStepPlan *plan)
{
    if (j->optim_kind < BRIEF_OPT_ADAMAX || j->optim_kind > BRIEF_OPT_SGD) return fail(BRIEF_ERR_INVALID, "bad optimizer kind");
    if (j->optim_kind != BRIEF_OPT_SGD && (!j->state1 || !j->state2)) return fail(BRIEF_ERR_INVALID, "optimizer state required");
    UpdatePayload up;
    up.opt = optim_scalars(j->optim_kind, j->lr, j->beta1, j->beta2, j->eps, j->t0 + 1);      // (replaced per step)
    up.params = j->params; up.s1 = j->state1; up.s2 = j->state2; up.pk = j->packed;
    return train_step_impl(&j->desc, j->packed, &j->grid, &j->batch, j->loss_kind, j->thr, j->beta, j->grads, j->loss_out, nullptr,
                           j->workspace, j->workspace_bytes, nullptr, &up, plan);
}

// tables of a group -> the first job's workspace (one tiny launch per job: the values travel as kernel arguments, no host buffer has to
// outlive the call), and the static halves of the two launches' arguments
static int fit_group_init(FitGroup &g, const brief_fit_job *jobs, hipStream_t st)
{
    const brief_fit_job &j0 = jobs[g.jobs[0]];
    const WsLayout wl0 = ws_layout(j0.desc, j0.batch.n);
    char *base = (char *)((float *)j0.workspace + wl0.table);
    g.fa_table = (FusedArgs *)base;
    g.ra_table = (ReduceArgs *)(base + ((size_t)BRIEF_GROUP_MAX * sizeof(FusedArgs) + 15) / 16 * 16);
    memset(&g.sg, 0, sizeof(g.sg));
    memset(&g.rg, 0, sizeof(g.rg));
    g.sg.table = g.fa_table; g.sg.njobs = g.njobs;
    g.rg.table = g.ra_table; g.rg.njobs = g.njobs;
    int wg = 0, rb = 0;
    for (int i = 0; i < g.njobs; ++i) {
        SmallStepPlan p;
        if (int rc = fit_job_plan(&jobs[g.jobs[i]], &p)) return rc;
        if (p.nt != g.nt || p.hb != g.hb) return fail(BRIEF_ERR_INVALID, "internal: mixed kernel variants in one group");
        hipLaunchKernelGGL(k_put<FusedArgs>, dim3(1), dim3(64), 0, st, g.fa_table + i, p.fa);
        hipLaunchKernelGGL(k_put<ReduceArgs>, dim3(1), dim3(64), 0, st, g.ra_table + i, p.ra);
        g.sg.wg_begin[i] = wg; wg += p.grid1;
        g.rg.blk_begin[i] = rb; rb += p.nb_reduce;
        g.rg.nb_hidden[i] = p.nb_hidden;
    }
    g.sg.wg_begin[g.njobs] = wg; g.rg.blk_begin[g.njobs] = rb;
    g.small_grid_total = wg; g.reduce_grid_total = rb;
    HIP_TRY(hipGetLastError());
    return 0;
}

// step number k of this call for every job of the group: two launches
static int fit_group_step(FitGroup &g, const brief_fit_job *jobs, int64_t k, double *lrs, hipStream_t st)
{
    for (int i = 0; i < g.njobs; ++i) {
        const brief_fit_job *j = &jobs[g.jobs[i]];
        const int64_t t = j->t0 + 1 + k;
        fit_job_lr(j, t, k, &lrs[g.jobs[i]]);
        g.sg.rng_step[i] = (uint64_t)t;
        g.sg.idx[i] = j->batch.idx ? j->batch.idx + k * j->idx_stride : nullptr;
        g.rg.opt[i] = optim_scalars(j->optim_kind, lrs[g.jobs[i]], j->beta1_table ? j->beta1_table[k] : j->beta1, j->beta2, j->eps, t);
        g.rg.loss_out[i] = j->loss_log ? j->loss_log + k : j->loss_out;
    }
#define BRIEF_CASE(NTV, HBV)                                                                                        \
    if (g.nt == NTV && g.hb == HBV)                                                                                 \
        hipLaunchKernelGGL((k_small_group<NTV, HBV>), dim3(g.small_grid_total), dim3(256), sizeof(float) * SmallLds<NTV>::TOTAL, st, g.sg);
    BRIEF_CASE(1, 1) BRIEF_CASE(1, 3) BRIEF_CASE(1, 5) BRIEF_CASE(1, 7)
    BRIEF_CASE(2, 1) BRIEF_CASE(2, 3) BRIEF_CASE(2, 5) BRIEF_CASE(2, 7)
#undef BRIEF_CASE
    hipLaunchKernelGGL(k_reduce_group, dim3(g.reduce_grid_total), dim3(256), 0, st, g.rg);
    HIP_TRY(hipGetLastError());
    return 0;
}


int brief_multi_fit(const brief_fit_job *jobs, int32_t njobs, int64_t steps, void *stream)
{
    if (njobs < 1 || !jobs) return fail(BRIEF_ERR_INVALID, "no jobs");
    if (steps < 0) return fail(BRIEF_ERR_INVALID, "bad step count");
    if (njobs > 4096) return fail(BRIEF_ERR_INVALID, "too many jobs for one call");
    for (int j = 0; j < njobs; ++j)
        if (int rc = fit_job_check(&jobs[j])) return rc;
    if (njobs == 1) return brief_siren_fit(jobs, steps, stream);
    std::lock_guard<std::recursive_mutex> state_lock(g_state_mu);
    DevState *ds = dev_state();      // the stream / event pool of the CURRENT device (created on first use there)
    if (!ds) return fail(BRIEF_ERR_INVALID, "out of host memory");
    if (!ds->pool_init) {
        for (int s = 0; s < kPoolStreams; ++s) HIP_TRY(hipStreamCreateWithFlags(&ds->pool[s], hipStreamNonBlocking));
        for (int s = 0; s <= kPoolStreams; ++s) HIP_TRY(hipEventCreateWithFlags(&ds->pool_ev[s], hipEventDisableTiming));
        ds->pool_init = true;
    }
    hipStream_t *g_pool = ds->pool;
    hipEvent_t *g_pool_ev = ds->pool_ev;
    // Units of work: GROUPS of narrow nets of one kernel variant (k_small_group: one launch per step for up to BRIEF_GROUP_MAX jobs —
    // the many small blocks of a DivideTask, where a launch pair per block and step left the GPU waiting for the host) and single
    // jobs (everything else: their own launches).  Unit u runs on internal stream u mod 8.
    FitGroup *groups = (FitGroup *)malloc(sizeof(FitGroup) * (size_t)njobs);
    int *unit_of = (int *)malloc(sizeof(int) * (size_t)njobs);          // job -> unit (group id, or -1 - single index)
    double *lrs = (double *)malloc(sizeof(double) * (size_t)njobs);
    if (!groups || !unit_of || !lrs) { free(groups); free(unit_of); free(lrs); return fail(BRIEF_ERR_INVALID, "out of host memory"); }
    int ngroups = 0;
    for (int j = 0; j < njobs; ++j) {
        lrs[j] = jobs[j].lr;
        unit_of[j] = -1;
        if (!use_small(jobs[j].desc) || prof_live()) continue;      // (timed runs keep their per-job launches)
        const int nt = brief_nt(jobs[j].desc), hb = small_hb(jobs[j].desc);
        int gi = -1;
        for (int q = ngroups - 1; q >= 0; --q)
            if (groups[q].nt == nt && groups[q].hb == hb) { if (groups[q].njobs < BRIEF_GROUP_MAX) gi = q; break; }
        if (gi < 0) { gi = ngroups++; groups[gi].nt = nt; groups[gi].hb = hb; groups[gi].njobs = 0; }
        groups[gi].jobs[groups[gi].njobs++] = j;
        unit_of[j] = gi;
    }
    for (int q = 0; q < ngroups; ++q)
        if (groups[q].njobs == 1) { unit_of[groups[q].jobs[0]] = -1; groups[q].njobs = 0; }      // a group of one: plain launches
    // stream slots: groups first, then the single jobs
    int nunits = 0;
    int *gslot = (int *)malloc(sizeof(int) * (size_t)(ngroups + 1));
    if (!gslot) { free(groups); free(unit_of); free(lrs); return fail(BRIEF_ERR_INVALID, "out of host memory"); }
    for (int q = 0; q < ngroups; ++q) gslot[q] = groups[q].njobs ? nunits++ : -1;
    for (int j = 0; j < njobs; ++j)
        if (unit_of[j] < 0) unit_of[j] = -1 - (nunits++);
    const int ns = nunits < kPoolStreams ? nunits : kPoolStreams;
    hipStream_t caller = (hipStream_t)stream;
    int rc = 0;
    // fork: everything already queued on the caller's stream happens before the first step of every job
    if (hipEventRecord(g_pool_ev[kPoolStreams], caller) != hipSuccess) rc = fail(BRIEF_ERR_LAUNCH, "hipEventRecord");
    for (int s = 0; s < ns && !rc; ++s)
        if (hipStreamWaitEvent(g_pool[s], g_pool_ev[kPoolStreams], 0) != hipSuccess) rc = fail(BRIEF_ERR_LAUNCH, "hipStreamWaitEvent");
    for (int q = 0; q < ngroups && !rc; ++q)
        if (groups[q].njobs) rc = fit_group_init(groups[q], jobs, g_pool[gslot[q] % ns]);
    // step-major order: the host feeds all streams evenly instead of running ahead on one of them
    for (int64_t k = 0; k < steps && !rc; ++k) {
        for (int q = 0; q < ngroups && !rc; ++q)
            if (groups[q].njobs) rc = fit_group_step(groups[q], jobs, k, lrs, g_pool[gslot[q] % ns]);
        for (int j = 0; j < njobs && !rc; ++j)
            if (unit_of[j] < 0) rc = fit_job_step(&jobs[j], jobs[j].t0 + 1 + k, k, &lrs[j], g_pool[(-1 - unit_of[j]) % ns]);
    }
    for (int j = 0; j < njobs && !rc && steps > 0; ++j) {
        hipStream_t sj = g_pool[(unit_of[j] < 0 ? -1 - unit_of[j] : gslot[unit_of[j]]) % ns];
        if (jobs[j].loss_log &&
            hipMemcpyAsync(jobs[j].loss_out, jobs[j].loss_log + steps - 1, sizeof(float), hipMemcpyDeviceToDevice, sj) != hipSuccess)
            rc = fail(BRIEF_ERR_LAUNCH, "hipMemcpyAsync");
    }
    free(groups); free(unit_of); free(lrs); free(gslot);
    // join (also on error: whatever was queued must be ordered before the caller's next work)
    for (int s = 0; s < ns; ++s) {
        HIP_TRY(hipEventRecord(g_pool_ev[s], g_pool[s]));
        HIP_TRY(hipStreamWaitEvent(caller, g_pool_ev[s], 0));
    }
    return rc;
}

int brief_optim_step(int kind, float *params, const float *grads, float *state1, float *state2, int64_t count,
                     double lr, double beta1, double beta2, double eps, int64_t t, void *stream)
{
    if (kind < BRIEF_OPT_ADAMAX || kind > BRIEF_OPT_SGD) return fail(BRIEF_ERR_INVALID, "bad optimizer kind");
    if (!params || !grads || count < 1 || t < 1) return fail(BRIEF_ERR_INVALID, "bad optimizer arguments");
    if (kind != BRIEF_OPT_SGD && (!state1 || !state2)) return fail(BRIEF_ERR_INVALID, "optimizer state required");
    const OptimScalars os = optim_scalars(kind, lr, beta1, beta2, eps, t);
    hipLaunchKernelGGL(k_optim, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, os, params, grads,
                       state1, state2, count);
    HIP_TRY(hipGetLastError());
    return 0;
}

int brief_profile_enable(int on)
{
    std::lock_guard<std::recursive_mutex> state_lock(g_state_mu);
    DevState *ds = dev_state();
    if (!ds) return fail(BRIEF_ERR_INVALID, "out of host memory");
    if (on && !ds->prof_init) {
        ds->prof_ev = (hipEvent_t *)calloc(2 * kProfSlots, sizeof(hipEvent_t));
        if (!ds->prof_ev) return fail(BRIEF_ERR_INVALID, "out of host memory");
        for (int i = 0; i < 2 * kProfSlots; ++i) HIP_TRY(hipEventCreate(&ds->prof_ev[i]));
        ds->prof_init = true;
    }
    g_prof_on = on != 0;
    g_prof_n = 0;
    g_prof_dev = current_device();
    return 0;
}

int brief_profile_fused(double *total_ms, int64_t *launches)
{
    if (!total_ms || !launches) return fail(BRIEF_ERR_INVALID, "null output");
    double tot = 0.0;
    DevState *ds = dev_state();
    if (g_prof_n > 0 && (g_prof_dev != current_device() || !ds || !ds->prof_init))
        return fail(BRIEF_ERR_INVALID, "brief_profile_fused must be called on the device brief_profile_enable was called on");
    for (int i = 0; i < g_prof_n; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(ds->prof_ev[2 * i + 1]));
        HIP_TRY(hipEventElapsedTime(&ms, ds->prof_ev[2 * i], ds->prof_ev[2 * i + 1]));
        tot += ms;
    }
    *total_ms = tot;
    *launches = g_prof_n;
    return 0;
}

__global__ void k_sincos_probe(const float *__restrict__ x, float *__restrict__ s, float *__restrict__ c, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float sv, cv;
    brief_fast_sincosf(x[i], &sv, &cv);
    s[i] = sv; c[i] = cv;
}

int brief_sincos_probe(const float *x, float *s, float *c, int64_t n, void *stream)
{
    if (!x || !s || !c || n < 1) return fail(BRIEF_ERR_INVALID, "bad probe arguments");
    hipLaunchKernelGGL(k_sincos_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, s, c, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

int brief_cu_count(void) { return cu_count(); }

int brief_deblock_edge(uint16_t *img, int64_t D, int64_t H, int64_t W, int z1, int z2, int fixed, int a1, int a2, int vertical,
                       double index_a, double index_b, double thres, int mode, void *stream)
{
    if (!img || z1 < 0 || z2 >= D || z2 < z1 || a2 < a1 || a1 < 0) return fail(BRIEF_ERR_INVALID, "bad deblock edge");
    const int64_t lim = vertical ? W : H, len = vertical ? H : W;
    if (a2 >= len || fixed < 0 || fixed >= lim) return fail(BRIEF_ERR_INVALID, "deblock edge outside the volume");
    if (fixed - 3 < 0 || fixed + 3 > lim - 1) return 0;      // deblock.py:57-63 / deblock.cpp:283-286: too close to the border
    const double alpha = 0.8 * (pow(2.0, (mode == 0 ? (double)(float)index_a : index_a) / 6.0) - 1.0);
    const double beta = 0.5 * (mode == 0 ? (double)(float)index_b : index_b) - 7.0;
    const int nz = z2 - z1 + 1, na = a2 - a1 + 1;
    const int64_t total = (int64_t)nz * na;
    hipLaunchKernelGGL(k_deblock_edge, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img, H, W, z1, nz,
                       fixed, a1, na, vertical, alpha, beta, thres, mode);
    HIP_TRY(hipGetLastError());
    return 0;
}

int brief_ssim_u16(const uint16_t *a, const uint16_t *b, int64_t D, int64_t H, int64_t W, const float *window11, double data_range,
                   double *partial, int64_t partial_count, void *stream)
{
    if (!a || !b || !window11 || !partial || D < 1) return fail(BRIEF_ERR_INVALID, "bad ssim arguments");
    if (H < 11 || W < 11) return fail(BRIEF_ERR_INVALID, "ssim needs H, W >= 11 (the reference skips the blur on shorter axes)");
    const int tiles_h = (int)((H - 10 + SSIM_TH - 1) / SSIM_TH), tiles_w = (int)((W - 10 + SSIM_TW - 1) / SSIM_TW);
    const int64_t blocks = D * tiles_h * tiles_w;
    if (partial_count < blocks || blocks > 0x7fffffff) return fail(BRIEF_ERR_WORKSPACE, "ssim partial buffer too small");
    const float C1 = (float)((0.01 * data_range) * (0.01 * data_range)), C2 = (float)((0.03 * data_range) * (0.03 * data_range));
    hipLaunchKernelGGL(k_ssim_u16, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, H, W, tiles_h, tiles_w, window11, C1, C2, partial);
    HIP_TRY(hipGetLastError());
    return 0;
}

int64_t brief_ssim_partial_count(int64_t D, int64_t H, int64_t W)
{
    if (H < 11 || W < 11 || D < 1) return -1;
    return D * ((H - 10 + SSIM_TH - 1) / SSIM_TH) * ((W - 10 + SSIM_TW - 1) / SSIM_TW);
}

int brief_sample_indices(int64_t *idx, int64_t n, int64_t pop, uint64_t seed, uint64_t step, void *stream)
{
    if (!idx || n < 1 || pop < 1) return fail(BRIEF_ERR_INVALID, "bad sampler arguments");
    hipLaunchKernelGGL(k_sample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx, n, (uint64_t)pop, seed, step);
    HIP_TRY(hipGetLastError());
    return 0;
}

int brief_sse_u16(const uint16_t *a, const uint16_t *b, int64_t n, double *sse_out, void *stream)
{
    if (!a || !b || !sse_out || n < 1) return fail(BRIEF_ERR_INVALID, "bad sse arguments");
    hipStream_t st = (hipStream_t)stream;
    // the first 8 bytes of the output double are used as the integer accumulator, then converted in place
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(sse_out);
    HIP_TRY(hipMemsetAsync(acc, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_sse_u16, dim3(1024), dim3(256), 0, st, a, b, n, acc);
    hipLaunchKernelGGL(k_u64_to_double, dim3(1), dim3(1), 0, st, acc, sse_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

}   // extern "C"
