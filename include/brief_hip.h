/* brief_hip.h — C-ABI of libbrief_hip.so: the MI355X-native SIREN fit/decode hot path of BRIEF.
 *
 * The reference (RichealYoung/BRIEF_PyTorch) has no FFI for this path: it sits behind the
 * Python object returned by init_phi() (utils/Networks.py:795-802) and the loop body in
 * main.py:385-400.  Each entry point below names the reference code it replaces.  All pointers
 * are DEVICE pointers (e.g. torch.Tensor.data_ptr()), `stream` is a hipStream_t passed as
 * void*.  The compute entry points only enqueue kernels on `stream`: no device memory is allocated or freed and the
 * host never waits (exceptions, all outside the per-step path: brief_multi_fit creates its internal stream pool on
 * first use and forks / joins it with events; brief_profile_* record and wait for events; brief_sse_u16 issues a
 * hipMemsetAsync).  Every function returns 0 on success or a negative brief_status; the message of the
 * last failure on the calling thread is available through brief_last_error().
 */
#ifndef BRIEF_HIP_H
#define BRIEF_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BRIEF_VERSION 130 /* 0.1.3: features up to 4096 (k_wide; brief_siren_forward_ws + brief_forward_workspace_bytes: inference above 1024 features needs a
                           * scratch), library state per device; struct layouts as in 0.1.1 (brief_fit_job with lr_table, beta1_table, idx_stride) */

typedef enum {
    BRIEF_OK = 0,
    BRIEF_ERR_INVALID = -1,      /* bad argument / unsupported configuration (e.g. res=True, F > 4096) */
    BRIEF_ERR_LAUNCH = -2,       /* HIP launch / runtime error */
    BRIEF_ERR_WORKSPACE = -3     /* workspace too small */
} brief_status;

/* SIREN(coords_channel, data_channel, features, layers, w0, output_act)  utils/Networks.py:246-266.
 * layers = number of Linear layers; hidden Sine() is hard-coded w0=30 in the reference (:228,259). */
typedef struct {
    int32_t cin;         /* 2 | 3 */
    int32_t cout;        /* 1 .. 4 */
    int32_t layers;      /* >= 2 */
    int32_t features;    /* 1 .. 4096 for BRIEF_PREC_F32 (padded internally to whole 32-feature tiles; SIREN.calc_features, utils/Networks.py:299-314,
                          * has no width limit: the shipped default.yaml solves to 527 on a 512^3 uint16 volume and to 1494 on a 1024^3 one), 1 .. 512 for BRIEF_PREC_BF16,
                          * 1 .. 256 for BRIEF_PREC_BF16X3 */
    float w0_first;
    float w0_hidden;
    int32_t output_act;
    int32_t precision;   /* BRIEF_PREC_F32 (0): exact f32 MFMA everywhere.  BRIEF_PREC_BF16 (1): the hidden F x F GEMMs run on the
                          * bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, f32 accumulate, f32 master weights, f32 first layer, head,
                          * loss, reductions and optimizer); activations/deltas are stashed as bf16.  The MI355X counterpart of the
                          * reference's low-precision mode (Compress.half, main.py:388-399), pinned by a PSNR band, not bitwise.
                          * BRIEF_PREC_BF16X3 (2), features <= 256: split precision — weights, activations and deltas of the hidden GEMMs
                          * are split into hi + lo 16-bit halves and every product is three 16-bit MFMAs (hi.hi + hi.lo + lo.hi, f32
                          * accumulate): fp16 halves in the forward chains (22 significant bits), bf16 halves in the backward chains and
                          * the weight-gradient GEMM (16 bits, bf16's exponent range).  Held to the SAME oracle bands as BRIEF_PREC_F32
                          * (forward 2e-5, gradients 1e-4, traces 1e-4) but not bit-identical to it; never the default.  Inference
                          * (brief_siren_forward / _decode) runs the same fp16-halves forward chains (~1e-6 from the f32 kernel).  Needs |w0 W / 2 pi| < 1000 for every hidden weight (fp16 range of the scaled forward copy). */
} brief_siren_desc;
enum { BRIEF_PREC_F32 = 0, BRIEF_PREC_BF16 = 1, BRIEF_PREC_BF16X3 = 2 };

/* create_flattened_coords(shape, mode)  utils/dataset.py:36-60: linspace(lo,hi,n) per axis, (d,h,w) order */
typedef struct {
    int32_t ndim;        /* == cin */
    int32_t reserved;
    int64_t dims[3];
    float lo, hi;
} brief_grid_desc;

/* where the samples of one step come from.
 *   j = idx ? idx[n] : (rng_pop > 0 ? philox(rng_seed, rng_step, n) mod-free-scaled to [0, rng_pop) : n + offset)
 *                                            (RandompointSampler main.py:154-163 / full-volume cube :112-125;
 *                                             the in-kernel stream equals brief_sample_indices(.., rng_pop, rng_seed, rng_step))
 *   x = coords ? coords[j, :] : grid coordinate of voxel j
 *   y = targets[j, :]      w = weights ? weights[j, :] : 1 */
typedef struct {
    const float *coords;     /* [*, cin] or NULL */
    const float *targets;    /* [*, cout]  (ignored by forward) */
    const float *weights;    /* [*, cout] or NULL */
    const int64_t *idx;      /* [n] or NULL */
    int64_t offset;
    int64_t n;               /* samples in this call */
    int64_t rng_pop;         /* > 0 with idx == NULL: draw the voxel indices in-kernel (train step only) */
    uint64_t rng_seed, rng_step;
} brief_batch_desc;

typedef enum {
    BRIEF_LOSS_L2 = 0, BRIEF_LOSS_SMOOTHL1 = 1,      /* main.py:176-191 */
    BRIEF_LOSS_EXTERNAL = 2   /* `targets` holds dL/dyhat per sample and channel (what autograd hands to the module's backward,
                               * main.py:396 `loss.backward()`): it is used as is (no weights, no 1/N), loss_out receives 0 */
} brief_loss_kind;
typedef enum { BRIEF_OPT_ADAMAX = 0, BRIEF_OPT_ADAM = 1, BRIEF_OPT_SGD = 2 } brief_optim_kind; /* utils/misc.py:174-183 */
typedef enum { BRIEF_OUT_F32 = 0, BRIEF_OUT_U8 = 1, BRIEF_OUT_U16 = 2 } brief_out_kind;

int brief_version(void);
const char *brief_last_error(void);

/* number of floats in the canonical packed parameter buffer (W0,b0,W1,b1,... == torch parameters()
 * order; W_l row-major [out,in] as in utils/ModelSave.py:32-50).  == SIREN.calc_param_count */
int64_t brief_param_count(const brief_siren_desc *d);
/* floats in the derived MFMA-fragment-ordered weight buffer produced by brief_siren_repack */
int64_t brief_packed_count(const brief_siren_desc *d);
/* bytes of scratch a train step of n samples needs (activation stash + gradient slabs) */
int64_t brief_train_workspace_bytes(const brief_siren_desc *d, int64_t n);

/* canonical params -> fragment-ordered copies read by the kernels.  Call after every change of
 * `params` made outside brief_optim_step's caller loop (init, load_model utils/ModelSave.py:8-27). */
int brief_siren_repack(const brief_siren_desc *d, const float *params, float *packed, void *stream);

/* SIREN.forward under no_grad (main.py:266-268 sample_nf; utils/misc.py:59-92 reconstruct_flattened).
 * out_kind F32: out = yhat [n,cout] float.  U8/U16: fused invnormalize_data('minmaxany_a_b')
 * (utils/io.py:136-147): clip((yhat-a)/(b-a),0,1)*(vmax-vmin)+vmin truncated to the integer type. */
int brief_siren_forward(const brief_siren_desc *d, const float *packed, const brief_grid_desc *grid,
                        const brief_batch_desc *batch, void *out, int out_kind,
                        float scale_min, float scale_max, double vmin, double vmax, void *stream);
/* The same with a caller-provided scratch.  Nets of more than 1024 features keep no layer in LDS whole: their activations travel
 * through two ping-pong planes per workgroup in `workspace` (brief_forward_workspace_bytes(d, n) bytes, 0 up to 1024 features, where
 * workspace may be NULL); brief_siren_forward refuses such a net with BRIEF_ERR_WORKSPACE. */
int64_t brief_forward_workspace_bytes(const brief_siren_desc *d, int64_t n);
int brief_siren_forward_ws(const brief_siren_desc *d, const float *packed, const brief_grid_desc *grid,
                           const brief_batch_desc *batch, void *out, int out_kind,
                           float scale_min, float scale_max, double vmin, double vmax, void *workspace, int64_t workspace_bytes, void *stream);

/* zero_grad + forward + loss + backward of main.py:385-396 for one batch.
 * grads: canonical packed layout, fully overwritten.  loss_out: one float (mean loss).
 * yhat_out: optional [n,cout].  thr: normalised weight_thres (0 disables, main.py:178-179). */
int brief_siren_train_step(const brief_siren_desc *d, const float *packed, const brief_grid_desc *grid,
                           const brief_batch_desc *batch, int loss_kind, float thr, float beta,
                           float *grads, float *loss_out, float *yhat_out,
                           void *workspace, int64_t workspace_bytes, void *stream);

/* One whole iteration of the loop body main.py:385-400 in three launches: brief_siren_train_step followed by the
 * optimizer update of brief_optim_step, with the update applied inside the gradient reduction and written through
 * to BOTH the canonical `params` and the fragment-ordered `packed` copy (no separate repack).  `packed` must hold
 * a repack of `params` on entry.  Results are bit-identical to train_step + optim_step + repack. */
int brief_siren_fit_step(const brief_siren_desc *d, float *params, float *packed, const brief_grid_desc *grid,
                         const brief_batch_desc *batch, int loss_kind, float thr, float beta,
                         int optim_kind, float *state1, float *state2, double lr, double beta1, double beta2, double eps, int64_t t,
                         float *grads, float *loss_out, void *workspace, int64_t workspace_bytes, void *stream);

/* ---- many steps per call -------------------------------------------------------------------------
 * One independent fit (one block of a DivideTask partition, or the whole volume of a SingleTask): everything
 * the loop `for steps in range(1, max_steps + 1)` of main.py:385-402 touches, as plain device pointers.
 * `batch` is the per-step template: with idx == NULL and rng_pop > 0 the voxel indices of step t are drawn
 * in-kernel with rng_step = t (RandompointSampler, main.py:154-163); with rng_pop == 0 every step sees the
 * same batch (RandomCubeSampler at its default: the whole volume, main.py:112-125).  A per-step idx stream
 * cannot be expressed here: use brief_siren_fit_step for replayed indices.
 * lr schedule = torch MultiStepLR stepped after every optimizer step (utils/misc.py:184-197, main.py:400):
 * `lr` is the value in force for step t0 + 1; whenever the number of finished steps equals a milestone the
 * running value is multiplied by gamma (once per occurrence).  n_milestones == 0: constant lr. */
typedef struct brief_fit_job {
    brief_siren_desc desc;
    brief_grid_desc grid;
    brief_batch_desc batch;
    float *params, *packed;        /* canonical parameters and their fragment-ordered copy (repacked on entry) */
    float *state1, *state2;        /* optimizer state (exp_avg | exp_inf or exp_avg_sq); unused for SGD */
    float *grads;                  /* [param_count]: gradient of the last step */
    float *loss_out;               /* device scalar: loss of the last step */
    float *loss_log;               /* device [steps] or NULL: loss of every step of this call */
    void *workspace;               /* brief_train_workspace_bytes(desc, batch.n); private to this job */
    int64_t workspace_bytes;
    int32_t loss_kind, optim_kind;
    float thr, beta;
    double lr, beta1, beta2, eps;
    const int64_t *milestones;     /* host array, ascending */
    int32_t n_milestones, reserved;
    double gamma;
    int64_t t0;                    /* optimizer steps already taken; this call runs steps t0+1 .. t0+steps */
    /* closed-form schedules (StepLR, CyclicLR of utils/misc.py:184-197, which torch evaluates from the epoch count, not as a
     * running product): host arrays with one entry per step of THIS call, entry k for step t0+1+k.  lr_table overrides
     * lr / milestones / gamma; beta1_table (CyclicLR cycles the momentum of Adam-family optimizers) overrides beta1. */
    const double *lr_table, *beta1_table;
    /* batch.idx with idx_stride > 0: a device-resident index STREAM, step t0+1+k reads batch.idx + k * idx_stride (int64
     * elements).  How the windowed RandomCubeSampler (main.py:38-125) runs without a host round trip per step: the caller
     * expands the window draws of a run of steps into voxel indices once.  idx_stride == 0 with batch.idx set is refused. */
    int64_t idx_stride;
} brief_fit_job;

/* `steps` iterations of brief_siren_fit_step enqueued back to back on `stream` (3 launches each, no host
 * synchronisation, nothing allocated).  Bit-identical to calling brief_siren_fit_step `steps` times. */
int brief_siren_fit(const brief_fit_job *job, int64_t steps, void *stream);

/* Grouped independent fits (the per-block loop of main.py:547-575 for the blocks one GPU owns).  Narrow nets (features <= 64:
 * what BRIEF's own YAMLs produce) of one kernel variant are trained by ONE launch pair per step for up to 64 jobs (k_small_group +
 * k_reduce_group: workgroup ranges per job, the jobs' static arguments in a device table inside the first job's workspace) — the
 * many small blocks of a DivideTask no longer cost a launch pair per block and step; every other job runs its own launches.  Units
 * (groups and single jobs) run on internal HIP streams, unit u on stream u mod 8, forked from and joined back into `stream` with
 * events, so that kernels of different units overlap.  Each job's results are bit-identical to brief_siren_fit on its own; jobs
 * must not share any buffer except read-only targets/weights.  Call from one host thread per process (one process per GPU). */
int brief_multi_fit(const brief_fit_job *jobs, int32_t njobs, int64_t steps, void *stream);

/* optimizer.step() of main.py:399 (torch.optim.Adamax/Adam/SGD single-tensor rules); t is the
 * 1-based step count, lr the scheduler's current value.  state1/state2: exp_avg / exp_inf|exp_avg_sq. */
int brief_optim_step(int kind, float *params, const float *grads, float *state1, float *state2, int64_t count,
                     double lr, double beta1, double beta2, double eps, int64_t t, void *stream);

/* uniform voxel indices in [0, pop) for one step: the device-side stand-in for
 * torch.randint(0, pop_size, (n,)) of main.py:156 (counter-based Philox4x32-10, keyed by seed/step). */
int brief_sample_indices(int64_t *idx, int64_t n, int64_t pop, uint64_t seed, uint64_t step, void *stream);

/* sum of squared differences of two integer volumes (for PSNR, utils/misc.py:451-456); sse_out: one double */
int brief_sse_u16(const uint16_t *a, const uint16_t *b, int64_t n, double *sse_out, void *stream);

/* cal_ssim of utils/misc.py:458-475 for single-channel uint16 volumes [D,H,W]: per z-slice 2-D SSIM (utils/ssim.py:
 * 11-tap Gaussian `window11`, valid padding, K=(0.01,0.03)).  Writes one double per 16x64 output tile, slice-major
 * (brief_ssim_partial_count of them; tiles of slice z are contiguous); slice mean = sum of its tiles / ((H-10)(W-10)),
 * SSIM = mean over slices.  Across ranks: all-reduce [sum of slice means, slices]. */
int64_t brief_ssim_partial_count(int64_t D, int64_t H, int64_t W);
int brief_ssim_u16(const uint16_t *a, const uint16_t *b, int64_t D, int64_t H, int64_t W, const float *window11, double data_range,
                   double *partial, int64_t partial_count, void *stream);

/* Block-boundary filter of DivideTask outputs (reference deblock.py:52-78 / deblock.cpp:277-319): filters the
 * boundary line  x == fixed, y in [a1,a2]  (vertical != 0)  or  y == fixed, x in [a1,a2]  (vertical == 0)  of
 * every slice z1..z2 of a uint16 volume [D,H,W] in place.  mode 1 = deblock.py arithmetic, 0 = deblock.cpp
 * arithmetic.  Lines must be issued in the reference's order (they overlap); see brief_pytorch_amd/deblock.py. */
int brief_deblock_edge(uint16_t *img, int64_t D, int64_t H, int64_t W, int z1, int z2, int fixed, int a1, int a2, int vertical,
                       double index_a, double index_b, double thres, int mode, void *stream);

/* measurement hooks (bench.py): while enabled, every train step records a HIP event pair on the
 * caller's stream around its dominant kernel (the fused forward/loss/dgrad launch);
 * brief_profile_fused waits for them and returns the summed duration and the launch count. */
int brief_profile_enable(int on);
int brief_profile_fused(double *total_ms, int64_t *launches);

/* diagnostics: the kernels' sine / cosine (two-term reduction to revolutions + v_sin_f32 / v_cos_f32, csrc/brief_math.h)
 * evaluated elementwise on device: s[i] = sin(x[i]), c[i] = cos(x[i]).  tests/test_sincos_host.py measures it against
 * float64 (Sine.forward of the reference goes through torch's ~1-ulp sin, utils/Networks.py:227-234). */
int brief_sincos_probe(const float *x, float *s, float *c, int64_t n, void *stream);

/* compute units of the current device as the library sized its grids and workspaces from (256 on a whole MI355X) */
int brief_cu_count(void);

#ifdef __cplusplus
}
#endif
#endif
