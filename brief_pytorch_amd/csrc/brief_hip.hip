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
#include "brief_layout.h"
#include "brief_math.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// row of accumulator register r (0..15) inside a 32x32 tile, for lane half hi
#define ROWMAP(r, hi) (((r) & 3) + 8 * ((r) >> 2) + 4 * (hi))
template <int V> struct RoleC { static constexpr int value = V; };

template <int NT, bool INFER = false>      // INFER: the inference kernels' wave mapping (brief_layout.h: brief_wm / brief_wm_infer)
struct KCfg {
    static constexpr int WM = INFER ? brief_wm_infer(NT) : brief_wm(NT);           // waves along the feature dimension
    static constexpr int WS = 4 / WM;                 // 32-sample tiles per workgroup
    static constexpr int MTW = (NT + WM - 1) / WM;    // feature tiles owned by one wave
    static constexpr bool EXACT = (NT % WM) == 0;
    static constexpr int FP = 32 * NT;
    static constexpr int XS_FLOATS = WS * NT * 1024;  // activation image(s)
    static constexpr int TROWS = 32 * MTW;            // local features one wave owns
    static constexpr int RS = (TROWS + 63) / 64;      // local-feature slots per lane in the skinny-gradient pass
    static constexpr int T_FLOATS = 4 * TROWS * 33;   // per-wave transposed scratch
    static constexpr int G_FLOATS = 4 * 256;          // per-wave g[4][32] + xs[32][4]
};

struct GridArgs {
    int ndim;
    int64_t dims[3];
    float lo, hi;
    float step[3];
    int fast;               // the grid has fewer than 2^32 points: 32-bit indices, division by multiplication
    uint64_t magic[3];      // floor(2^64 / dims[a]) + 1   (exact quotients for every 32-bit numerator)
};

// n / dv for a loop-invariant divisor (Lemire's fastdiv: one 64-bit multiply-high instead of ~100 VALU
// instructions of 64-bit division; the VALU is what the f32 MFMA competes with)
__device__ __forceinline__ uint32_t fast_div(uint32_t nn, uint64_t magic, uint32_t dv)
{
    return dv == 1 ? nn : (uint32_t)__umul64hi(magic, (uint64_t)nn);
}

struct FusedArgs {
    brief_siren_desc d;
    const float *pk;
    const float *coords;
    const float *targets;
    const float *weights;
    const int64_t *idx;
    int64_t offset;
    int64_t n;
    uint64_t rng_pop, rng_seed, rng_step;   // idx == NULL && rng_pop > 0: sample j in-kernel (Philox)
    GridArgs grid;
    int loss_kind;
    float thr, beta, inv_count;
    float *Z;            // [(L-2)][npad/32][FP][32]   phases (om z reduced to revolutions) of layers 0..L-3, tile-blocked: a 32-sample tile's rows are 128 B apart
    float *D;            // [(L-2)][npad/32][FP][32]   deltas of layers 1..L-2, same layout
    int64_t npad;
    float *rec;          // [gridDim.x*4][BRIEF_REC_FLOATS]
    int64_t n_begin, n_end;   // bf16 path: the sample range of this launch (body / tail launches)
    int rec_base;             // bf16 path: first record slot of this launch
    void *S16[5];        // bf16 path: stashes THETA (fp16 phases) | unused | D (bf16 deltas) ([(L-1)][FP][npad] each), X [4][npad], G [4][npad]
    float *slabs;        // k_small only: [(L-2)][gridDim.x][FP*FP + FP] per-workgroup hidden-layer gradient partials
    float *yhat_out;     // [n][cout] or NULL
    void *out;           // forward output
    int out_kind;
    float scale_min, den, span, vmin;   // fused invnormalize
    int stagger_cus;     // workgroups per residency slot (= CU count)
    int stagger;         // s_sleep(127) units of start delay per slot
    int pers_wgs;        // k_fused: workgroups [0, pers_wgs) walk tiles b, b + pers_wgs, ... < pers_tiles (the persistent body);
    int64_t pers_tiles;  //          workgroup b >= pers_wgs takes the single tile pers_tiles + b - pers_wgs (see fused_plan)
    int diag;            // timing diagnostics only (BRIEF_DIAG): bit 0 = stash descriptors with zero records (the range check then
                         // drops every stash load and store: results are wrong, the instruction stream is unchanged)
};

// Workgroup barrier for LDS hand-offs only.  __syncthreads() also emits s_waitcnt vmcnt(0), which
// drains every global load / store still in flight (register prefetches, stash stores) at each
// barrier; all cross-wave traffic in these kernels goes through LDS, so waiting for the LDS queue is
// enough (cdna_hip_programming.md, 'Pipelining across barriers').
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// torch.linspace as torch-CPU evaluates it (utils/dataset.py:28-32)
__device__ __forceinline__ float lin_coord(const GridArgs &g, int a, int64_t i)
{
    const int64_t nn = g.dims[a];
    if (nn == 1) return g.lo;
    return i < nn / 2 ? __fmaf_rn(g.step[a], (float)i, g.lo) : __fmaf_rn(-g.step[a], (float)(nn - 1 - i), g.hi);
}

__device__ __forceinline__ float lin_coord32(const GridArgs &g, int a, uint32_t i)
{
    const uint32_t nn = (uint32_t)g.dims[a];
    if (nn == 1) return g.lo;
    return i < nn / 2 ? __fmaf_rn(g.step[a], (float)i, g.lo) : __fmaf_rn(-g.step[a], (float)(nn - 1 - i), g.hi);
}

// voxel index -> coordinates of the flattened (d,h,w) / (h,w) grid (create_flattened_coords, utils/dataset.py:36-62)
__device__ __forceinline__ void grid_coords(const GridArgs &g, int cin, int64_t j, float &x0, float &x1, float &x2)
{
    if (cin == 3) {
        int64_t id, ih, iw;
        if (g.fast) {
            const uint32_t ju = (uint32_t)j, d2 = (uint32_t)g.dims[2], d1 = (uint32_t)g.dims[1];
            const uint32_t t2 = fast_div(ju, g.magic[2], d2);
            const uint32_t t1 = fast_div(t2, g.magic[1], d1);
            x0 = lin_coord32(g, 0, t1);
            x1 = lin_coord32(g, 1, t2 - t1 * d1);
            x2 = lin_coord32(g, 2, ju - t2 * d2);
            return;
        } else {
            iw = j % g.dims[2];
            const int64_t t2 = j / g.dims[2];
            ih = t2 % g.dims[1]; id = t2 / g.dims[1];
        }
        x0 = lin_coord(g, 0, id);
        x1 = lin_coord(g, 1, ih);
        x2 = lin_coord(g, 2, iw);
    } else {
        int64_t ih, iw;
        if (g.fast) {
            const uint32_t ju = (uint32_t)j, d1 = (uint32_t)g.dims[1];
            const uint32_t t1 = fast_div(ju, g.magic[1], d1);
            x0 = lin_coord32(g, 0, t1);
            x1 = lin_coord32(g, 1, ju - t1 * d1);
            return;
        } else {
            iw = j % g.dims[1]; ih = j / g.dims[1];
        }
        x0 = lin_coord(g, 0, ih);
        x1 = lin_coord(g, 1, iw);
    }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 16-byte buffer load: one VGPR byte offset per lane + a scalar byte offset, so no per-block
// 64-bit address VGPRs exist for the compiler to hoist and spill (cdna_hip_programming.md T8).
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
#ifndef BRIEF_WGRAD_AUX
#define BRIEF_WGRAD_AUX 0      // cache policy of k_wgrad's operand panels (2: streaming)
#endif
__device__ __forceinline__ float4 bload4w(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, BRIEF_WGRAD_AUX);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
__device__ __forceinline__ void bstore1(float v, __amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, voff, soff, 0);
}
// the split-precision fused kernel's stash accesses, with a cache-policy knob of their own (see BRIEF_X3W_AUX for the reader's side)
#ifndef BRIEF_STASH_LD_AUX
#define BRIEF_STASH_LD_AUX 0      // phase reloads of the dgrad chains (k_fused and k_fused_x3): read once (experiment: 2)
#endif
#ifndef BRIEF_X3_STASH_AUX
#define BRIEF_X3_STASH_AUX 0      // k_fused_x3's own stash stores / phase reloads: the streaming policy costs it 9-10 us (measured), default policy
#endif
__device__ __forceinline__ float bload1s(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, BRIEF_STASH_LD_AUX));
}
__device__ __forceinline__ void bstore1s(float v, __amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, voff, soff, BRIEF_X3_STASH_AUX);
}

// One layer's GEMM for the feature tiles this wave owns:  acc[t] += A(mt,:) * image.
// A fragments stream from the packed weight buffer (L2 resident) PD (kt,q)-steps ahead of use.
// kit: (k-tile, q) steps that hold real features, ceil(F / 8): the steps above it multiply zero weight columns with the
// zero activations of the padding features, so skipping them changes nothing but the time (F = 22 runs 3 of its 4 steps)
template <int NT, bool INFER = false>
__device__ __forceinline__ void chain(f32x16 (&acc)[(KCfg<NT, INFER>::MTW)], __amdgpu_buffer_rsrc_t rs, int soff_layer /*bytes*/,
                                      const float4 *Xs, int wm, int lane, int kit)
{
    using K = KCfg<NT, INFER>;
    constexpr int NIT = NT * 4;
    constexpr int PD = NIT < 4 ? NIT : 4;
    const int voff = lane * 16;
    int soff_w = soff_layer + wm * (NT * 4 * 1024);
    // opaque to the optimiser: otherwise all 2*NT*4 per-block scalar offsets are hoisted out of the tile
    // loop, spilled to VGPR lanes and fetched back with v_readlane (a VALU op the f32 MFMA has to wait for)
    asm volatile("" : "+s"(soff_w));
    constexpr int smul = 1024;
    float4 areg[NIT][K::MTW];
    float4 breg[NIT];
    // padding is at most 31 features = the last three steps: only those carry a run-time check (on their MFMAs)
#pragma unroll
    for (int it = 0; it < PD; ++it)
#pragma unroll
        for (int t = 0; t < K::MTW; ++t)
            if (K::EXACT || wm + K::WM * t < NT) areg[it][t] = bload4(rs, voff, soff_w + (K::WM * t * NT * 4 + it) * smul);
    breg[0] = Xs[lane];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        if (it >= NIT - 3 && it >= kit) break;
        // (the prefetch is not predicated on kit: behind a wave-uniform branch the s_waitcnt bookkeeping counts the load as possibly
        //  missing and waits one step early — prefetch depth 3, 2, 1 over steps NIT-7 .. NIT-5; the padded steps hold zero fragments)
        if (it + PD < NIT) {
#pragma unroll
            for (int t = 0; t < K::MTW; ++t)
                if (K::EXACT || wm + K::WM * t < NT) {
                    areg[it + PD][t] = bload4(rs, voff, soff_w + (K::WM * t * NT * 4 + it + PD) * smul);
                }
        }
        if (it + 1 < NIT) breg[it + 1] = Xs[(it + 1) * 64 + lane];
        // pin the prefetch above this step's MFMAs: left alone, the scheduler sinks each load to
        // just before its use (one register set, vmcnt(0) per step, L2 latency fully exposed)
        __builtin_amdgcn_sched_barrier(0);
        const float4 b = breg[it];
#pragma unroll
        for (int t = 0; t < K::MTW; ++t) {
            if (K::EXACT || wm + K::WM * t < NT) {
                const float4 a = areg[it][t];
                acc[t] = MFMA(a.x, b.x, acc[t]);
                acc[t] = MFMA(a.y, b.y, acc[t]);
                acc[t] = MFMA(a.z, b.z, acc[t]);
                acc[t] = MFMA(a.w, b.w, acc[t]);
            }
        }
    }
}

template <int NT, bool INFER = false>
__device__ __forceinline__ void write_image(float4 *Xs, const f32x16 (&h)[(KCfg<NT, INFER>::MTW)], int wm, int lane)
{
    using K = KCfg<NT, INFER>;
#pragma unroll
    for (int t = 0; t < K::MTW; ++t) {
        const int mt = wm + K::WM * t;
        if (K::EXACT || mt < NT) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                Xs[(mt * 4 + q) * 64 + lane] = make_float4(h[t][4 * q], h[t][4 * q + 1], h[t][4 * q + 2], h[t][4 * q + 3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// BRIEF_PREC_BF16X3: split-precision hidden GEMMs.  x = hi + lo with hi = bf16(x), lo = bf16(x - hi) keeps ~16 significant bits (fp16 halves, used by the forward chains: 22)
// of x; a product of two such operands is hi.hi + hi.lo + lo.hi (the dropped lo.lo term is 2^-16 of the product), three
// v_mfma_f32_32x32x16_bf16 with f32 accumulation where the exact path issues eight v_mfma_f32_32x32x2_f32: 96 instead of 512
// matrix-pipe cycles per 32 x 32 x 16 block.  tools/bf16x3_emulation.py: forward 7.9e-6 of max|y| and gradients 1.0e-5 of a
// tensor's max-abs against float64 on the 4x256 net (f32: 7e-7 / 4e-7; bands 2e-5 / 1e-4).  Same skeleton as the f32 path (32-sample
// tiles, f32 stashes, f32 head / loss / skinny gradients / optimizer); only the image and the weight fragments change.
typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 x3_f16x8 __attribute__((ext_vector_type(8)));
union X3Frag { uint4 u; x3_bf16x8 v; x3_f16x8 f; };
#define MFMA_X3(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define MFMA_X3F(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
// The FORWARD chains use fp16 halves (11 + 11 significant bits: the phase a deep net accumulates stays at f32's own accuracy,
// tools/fuzz_parity.py ... bf16x3): activations lie in [-1, 1] and the forward weight copies carry 2^6 on top of om / 2 pi so that
// their lo halves stay fp16-normal (|w om / 2 pi| ~ sqrt(6 / F) / 2 pi); the epilogue multiplies the accumulator by 2^-6, exactly.
// The BACKWARD chains and the weight-gradient GEMM keep bf16 halves: deltas span far more than fp16's 30 binades.
#define BRIEF_X3_FWD_SCALE 64.0f
#define BRIEF_X3_FWD_UNSCALE 0.015625f
__device__ __forceinline__ void x3_split_f16(float x, uint16_t &hi, uint16_t &lo)
{
    union { _Float16 h; uint16_t u; } a, b;
    a.h = (_Float16)x;
    b.h = (_Float16)(x - (float)a.h);
    hi = a.u; lo = b.u;
}
__device__ __forceinline__ void x3_split_bf16(float x, uint16_t &hi, uint16_t &lo)
{
    union { __bf16 h; uint16_t u; } a, b;
    a.h = (__bf16)x;
    b.h = (__bf16)(x - (float)a.h);
    hi = a.u; lo = b.u;
}

// registers 8s .. 8s+7 of an accumulator tile -> the hi and lo bf16 fragments of k-step s (the next layer's B operand)
template <bool F16>
__device__ __forceinline__ void x3_pack(const f32x16 &x, int s, uint4 &hi, uint4 &lo)
{
    X3Frag h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (F16) {
            const _Float16 b = (_Float16)x[8 * s + j];
            h.f[j] = b;
            l.f[j] = (_Float16)(x[8 * s + j] - (float)b);
        } else {
            const __bf16 b = (__bf16)x[8 * s + j];
            h.v[j] = b;
            l.v[j] = (__bf16)(x[8 * s + j] - (float)b);
        }
    }
    hi = h.u; lo = l.u;
}

// image layout: [hi | lo][kt][s][lane] x 16 B  (NT * 2 * 64 uint4 per half)
template <int NT, bool F16>
__device__ __forceinline__ void x3_write_image(uint4 *X16, const f32x16 (&h)[KCfg<NT>::MTW], int wm, int lane)
{
    using K = KCfg<NT>;
#pragma unroll
    for (int t = 0; t < K::MTW; ++t) {
        const int mt = wm + K::WM * t;
        if (K::EXACT || mt < NT) {
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                uint4 hi, lo;
                x3_pack<F16>(h[t], sx, hi, lo);
                X16[(mt * 2 + sx) * 64 + lane] = hi;
                X16[NT * 2 * 64 + (mt * 2 + sx) * 64 + lane] = lo;
            }
        }
    }
}

// one feature tile's 16 accumulator registers -> its two k-steps of hi | lo fragments in the image
template <bool F16>
__device__ __forceinline__ void x3_write_tile(uint4 *X16, const f32x16 &h, int mt, int lane)
{
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
        uint4 hi, lo;
        x3_pack<F16>(h, sx, hi, lo);
        X16[(mt * 2 + sx) * 64 + lane] = hi;
        X16[8 * 2 * 64 + (mt * 2 + sx) * 64 + lane] = lo;
    }
}

#ifndef BRIEF_X3_PD
#define BRIEF_X3_PD 2
#endif
// the first PD k-steps of a chain's A fragments, requested AHEAD of the stash traffic that precedes the chain: vector-memory
// operations retire in order (stores count in vmcnt too), so fragments requested after an epilogue's 32 stash stores (or the
// dgrad's 32 stores + 32 phase loads) make the chain's first MFMA wait for all of those
template <int NT>
struct X3Pre { uint4 hi[BRIEF_X3_PD][KCfg<NT>::MTW], lo[BRIEF_X3_PD][KCfg<NT>::MTW]; };

#define X3_LOAD_INTO(dh_, dl_, it_)                                                                          \
    _Pragma("unroll") for (int t = 0; t < K::MTW; ++t) {                                                     \
        if (K::EXACT || wm + K::WM * t < NT) {                                                               \
            const u32x4 vh = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff_w + (K::WM * t * NT * 2 + (it_)) * 1024, 0); \
            const u32x4 vl = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff_w + soff_lo + (K::WM * t * NT * 2 + (it_)) * 1024, 0); \
            dh_[t] = make_uint4(vh.x, vh.y, vh.z, vh.w);                                                     \
            dl_[t] = make_uint4(vl.x, vl.y, vl.z, vl.w);                                                     \
        }                                                                                                    \
    }

template <int NT>
__device__ __forceinline__ void x3_preload(X3Pre<NT> &pre, __amdgpu_buffer_rsrc_t rs, int soff_layer, int lo_bytes, int wm, int lane)
{
    using K = KCfg<NT>;
    const int voff = lane * 16;
    int soff_w = soff_layer + wm * (NT * 2 * 1024);
    asm volatile("" : "+s"(soff_w));
    int soff_lo = lo_bytes;
    asm volatile("" : "+s"(soff_lo));
#pragma unroll
    for (int it = 0; it < BRIEF_X3_PD; ++it) X3_LOAD_INTO(pre.hi[it], pre.lo[it], it)
}

#undef X3_LOAD_INTO

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 voxel-index stream (stands in for the CPU torch.randint of main.py:156)
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1)
{
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ int64_t philox_index(int64_t i, uint64_t pop, uint64_t seed, uint64_t step)
{
    uint32_t c[4] = {(uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)step, (uint32_t)(step >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const uint64_t r64 = ((uint64_t)c[0] << 32) | c[1];
    return (int64_t)__umul64hi(r64, pop);      // multiply-shift: bias < pop / 2^64
}

// LDS map of k_fused (floats):  R = max(image, transpose scratch) | G (per-wave g / coords) | HW (head weights)
//   image  X : WS x NT x 1024            activation register image(s)
//   scratch T: 4 waves x 32*MTW x 33      per-wave [feature][sample] transpose (aliases X: used only while
//                                         no wave reads the image, see the barriers below)
template <int NT, bool INFER = false>
struct FusedLds {
    using K = KCfg<NT, INFER>;
    static constexpr int R_FLOATS = K::XS_FLOATS > K::T_FLOATS ? K::XS_FLOATS : K::T_FLOATS;
    static constexpr int G_OFF = R_FLOATS;
    static constexpr int HW_OFF = G_OFF + K::G_FLOATS;
    static constexpr int TOTAL = HW_OFF + 4 * K::FP + 4;
};

#ifndef BRIEF_FUSED64
#ifndef BRIEF_LEAN_RM3
#define BRIEF_LEAN_RM3 1     // nt % 4 == 3: the three left-over tiles shared along K (lean_chain_rt3); 0: wave 3 computes a duplicate (A/B)
#endif
#define BRIEF_FUSED64 0      // 1: the 8-tile TRAIN step walks 64-sample tiles (k_lean<2, 2, 8>) instead of k_fused<8>'s 32-sample tiles
#endif
#ifndef BRIEF_KERNARG_RELOAD
#define BRIEF_KERNARG_RELOAD 1
#endif
#ifndef BRIEF_TRAIN_WPE
#define BRIEF_TRAIN_WPE 2   // waves per SIMD the TRAIN variant is register-allocated for
#endif
#if defined(BRIEF_STAMPS) && BRIEF_STAMPS + 0 != 2      /* -DBRIEF_STAMPS=2: only the start / end clock pair (the product instruction stream otherwise) */
#define STAMP(slot) { const long long t_ = clock64(); st_acc[slot] += (float)(t_ - st_last); st_last = t_; }
#else
#define STAMP(slot)
#endif
// resident TRAIN workgroups per CU: widths up to 4 tiles need <= 168 registers and run THREE (a third independent wave per SIMD
// fills what two leave idle: 5x128 0.325 -> 0.309 ms per step, 5x96 0.293 -> 0.267; tools/ab_wpe.py); 5-7 tiles need ~220-230
// registers: two (forced to 168 they spill ~40 registers and lose 1.6 %); above 8: one 512-register workgroup
// The 8-tile kernel (F = 225 ... 256, the headline) runs THREE workgroups per CU as well, in a lean form: it gives up the two
// register sets it parked across phases (the next layer's bias during an epilogue, the previous layer's phases during a dgrad
// chain: 64 registers; their latency is now covered by the two other waves of the SIMD) and fits 168 registers with one spilled
// dword: k_fused 0.680 -> 0.667 ms, 99.2 -> 100.1 M voxels/s (tools/ab_c2.sh).  5-7 tiles spill 36-46 registers in that form: two.
#ifndef BRIEF_LEAN
#define BRIEF_LEAN 1
#endif
#ifndef BRIEF_LEAN7
#define BRIEF_LEAN7 0      // 7: the 7-tile kernel in the lean three-workgroup form too (experiment: 168 VGPRs + 128 B of scratch)
#endif
constexpr bool fused_lean(int NT) { return BRIEF_LEAN && (NT == 8 || NT == BRIEF_LEAN7); }
constexpr int fused_train_wpe(int NT) { return NT > 8 ? 1 : (NT <= 4 || fused_lean(NT) ? (BRIEF_TRAIN_WPE > 3 ? BRIEF_TRAIN_WPE : 3) : BRIEF_TRAIN_WPE); }
template <int NT, bool TRAIN>
__global__ __launch_bounds__(256, TRAIN ? fused_train_wpe(NT) : (NT > 8 ? 2 : 3)) void k_fused(const FusedArgs a)
{
#ifdef BRIEF_STAMPS
    float st_acc[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    long long st_last = clock64();
    const long long st_c0 = st_last, st_r0 = wall_clock64();      // shader cycles / 100 MHz reference: the clock the kernel ran at
#endif
    using K = KCfg<NT, !TRAIN>;
    using LD = FusedLds<NT, !TRAIN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4 *X = reinterpret_cast<float4 *>(smem);
    float *T = smem;                       // aliases X
    float *G = smem + LD::G_OFF;
    float *HW = smem + LD::HW_OFF;         // Whp[4][FP], bhp[4]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, ln = lane & 31;
    const int wm = wave % K::WM, ws = wave / K::WM;
    const brief_siren_desc &d = a.d;
    const int L = d.layers, cin = d.cin, cout = d.cout;
    const int kit = (d.features + 7) >> 3;       // chain steps that hold real (unpadded) features
    const int64_t npad = a.npad;
    const float *pk = a.pk;
    float4 *Xs = X + ws * (NT * 256);
    float *Tw = T + wave * (K::TROWS * 33);
    float *Gw = G + wave * 256;
    const float4 *W0p = reinterpret_cast<const float4 *>(pk + brief_pk_w0(d));
    const __amdgpu_buffer_rsrc_t rs_pk =
        __builtin_amdgcn_make_buffer_rsrc((void *)pk, 0, (int)(brief_pk_count(d) * 4), 0x00020000);
    const int stash_bytes = (a.diag & 1) ? 0 : (int)((int64_t)K::FP * npad * 4);   // one layer's plane (host checks < 2^31)

    // head weights -> LDS once per workgroup (every lane needs all of them in the head dot product)
    {
        const float *headp = pk + brief_pk_head(d);
        for (int e = threadIdx.x; e < 4 * K::FP + 4; e += 256) HW[e] = headp[e];
    }
    // de-phase the co-resident workgroups: started together they run their MFMA chains and their
    // epilogues in lockstep and the matrix pipe idles through every epilogue
    {
        const int slot = (int)blockIdx.x < a.pers_wgs ? blockIdx.x / a.stagger_cus : 0;
        for (int i = 0; i < slot * a.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
    lds_barrier();

    // persistent skinny-gradient accumulators (lane <-> local feature)
    float acc0[K::RS][4], accWh[K::RS][4];
#pragma unroll
    for (int rs = 0; rs < K::RS; ++rs)
#pragma unroll
        for (int c = 0; c < 4; ++c) { acc0[rs][c] = 0.f; accWh[rs][c] = 0.f; }
    float accbh[4] = {0.f, 0.f, 0.f, 0.f};
    float lsum = 0.f;

    // Body (+ optional tail) in ONE launch.  The first pers_wgs workgroups are persistent and own a fixed set of tiles each;
    // workgroups above them (BRIEF_TAIL_ROUNDS > 0, a diagnostic: off by default) own ONE tile each and are dealt by the
    // hardware dispatcher to whichever CU retires a workgroup first — a dynamic queue without an atomic ticket, every
    // workgroup still sums a FIXED set of tiles into its record.  Measured (profiles/r03_wg_timeline.md): a tile's
    // LATENCY is 80 us alone on a CU and 95-104 us beside a mate whatever the plan, so the ragged end of the kernel is
    // about half a tile long under any tile-granular scheduling; the single-tile workgroups pay ~6 us per round for their
    // prologues and end 14-18 us later than the static plan, whose slot imbalance (first-dispatched workgroup of a CU: 6
    // tiles in 570 us, its mate: 625 us) happens to cancel against the 53 first-slot workgroups that carry a 7th tile.
    const bool pers = (int)blockIdx.x < a.pers_wgs;
    const int64_t tile_end = pers ? a.pers_tiles : a.pers_tiles + ((int64_t)blockIdx.x - a.pers_wgs) + 1;
    const int64_t tile_step = pers ? a.pers_wgs : 1;
    for (int64_t tile = pers ? (int64_t)blockIdx.x : a.pers_tiles + ((int64_t)blockIdx.x - a.pers_wgs); tile < tile_end; tile += tile_step) {
        const int64_t n0 = (tile * K::WS + ws) * 32;
        const int64_t n = n0 + ln;
        const bool valid = n < a.n;
        // ---- sample inputs (every wave of the sample tile fetches them; they are tiny).  Targets and
        //      weights are fetched now although the loss needs them a whole forward pass later.
        // The ~40 scalars this section needs (pointers, rng, grid dims and magics) are re-read from the kernarg
        // segment every tile through an opaque pointer: kept live across the tile loop they are spilled to VGPR
        // lanes and come back one v_readlane (a VALU instruction) at a time, ~450 per tile.
#if BRIEF_KERNARG_RELOAD
        typedef const __attribute__((address_space(4))) FusedArgs *kargs_f;
        kargs_f ap = (kargs_f)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ap));
        const int64_t *k_idx = ap->idx;
        const float *k_tg = ap->targets, *k_wt = ap->weights, *k_co = ap->coords;
        const uint64_t k_pop = ap->rng_pop, k_seed = ap->rng_seed, k_step = ap->rng_step;
        const int64_t k_off = ap->offset;
        GridArgs kg;
        kg.ndim = ap->grid.ndim; kg.lo = ap->grid.lo; kg.hi = ap->grid.hi; kg.fast = ap->grid.fast;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) { kg.dims[ax] = ap->grid.dims[ax]; kg.step[ax] = ap->grid.step[ax]; kg.magic[ax] = ap->grid.magic[ax]; }
#else
        const int64_t *k_idx = a.idx;
        const float *k_tg = a.targets, *k_wt = a.weights, *k_co = a.coords;
        const uint64_t k_pop = a.rng_pop, k_seed = a.rng_seed, k_step = a.rng_step;
        const int64_t k_off = a.offset;
        const GridArgs &kg = a.grid;
#endif
        int64_t j = 0;
        if (valid) j = k_idx ? k_idx[n] : (k_pop ? philox_index(n, k_pop, k_seed, k_step) : n + k_off);
        float x0 = 0.f, x1 = 0.f, x2 = 0.f;
        float yv[4] = {0.f, 0.f, 0.f, 0.f}, wv4[4] = {1.f, 1.f, 1.f, 1.f};
        if (valid) {
            if (TRAIN) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c < cout) {
                        yv[c] = k_tg[j * cout + c];
                        if (k_wt) wv4[c] = k_wt[j * cout + c];
                    }
                }
            }
            if (k_co) {
                x0 = k_co[j * cin];
                x1 = k_co[j * cin + 1];
                if (cin == 3) x2 = k_co[j * cin + 2];
            } else {
                grid_coords(kg, cin, j, x0, x1, x2);
            }
        }
        f32x16 acc[K::MTW];
        f32x16 creg[K::MTW];   // w*cos(w z) of the last sine layer (TRAIN)
        f32x16 hreg[K::MTW];
        float4 bnext[K::MTW][4];   // next hidden layer's bias, fetched one epilogue ahead of its use
#pragma unroll
        for (int t = 0; t < K::MTW; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { creg[t][r] = 0.f; hreg[t][r] = 0.f; }
#pragma unroll
            for (int q = 0; q < 4; ++q) bnext[t][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#define FUSED_LOAD_BIAS(layer)                                                                          \
    {                                                                                                   \
        const float *bp_ = pk + brief_pk_hidden(d, (layer)) + 2 * K::FP * K::FP;                        \
        _Pragma("unroll") for (int t = 0; t < K::MTW; ++t) {                                            \
            const int mt = wm + K::WM * t;                                                              \
            if (K::EXACT || mt < NT) {                                                                  \
                _Pragma("unroll") for (int q = 0; q < 4; ++q)                                           \
                    bnext[t][q] = *reinterpret_cast<const float4 *>(bp_ + 32 * mt + 8 * q + 4 * hi);    \
            }                                                                                           \
        }                                                                                               \
    }
        // ---- layer 0: z0 = W0 x + b0 as two K=2 MFMAs ([x0 x1 | x2 1] against W0p rows)
        {
            const float b0 = hi ? x1 : x0;
            const float b1 = hi ? 1.0f : x2;
#pragma unroll
            for (int t = 0; t < K::MTW; ++t) {
                const int mt = wm + K::WM * t;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
                if (K::EXACT || mt < NT) {
                    const float4 w = W0p[32 * mt + ln];
                    acc[t] = MFMA(hi ? w.y : w.x, b0, acc[t]);
                    acc[t] = MFMA(hi ? w.w : w.z, b1, acc[t]);
                }
            }
        }
        STAMP(0)
        // ---- sine layers 0 .. L-2
        for (int l = 0; l <= L - 2; ++l) {
            const bool last = (l == L - 2);
            if (l > 0) {
                if (TRAIN && fused_lean(NT)) FUSED_LOAD_BIAS(l)      // lean variant: no register set parked across the epilogue
#pragma unroll
                for (int t = 0; t < K::MTW; ++t) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[t][4 * q] = bnext[t][q].x; acc[t][4 * q + 1] = bnext[t][q].y;
                        acc[t][4 * q + 2] = bnext[t][q].z; acc[t][4 * q + 3] = bnext[t][q].w;
                    }
                }
                // matrix work first: the wave inside a chain outranks its SIMD mate's epilogue (-0.5 % step time; the opposite
                // order, epilogues first, costs +0.5 %: tools/ab_lib.sh, profiles/r02_issue_model.md)
                if (TRAIN) __builtin_amdgcn_s_setprio(3);
                chain<NT, !TRAIN>(acc, rs_pk, (int)(brief_pk_hidden(d, l) * 4), Xs, wm, lane, kit);
                if (TRAIN) __builtin_amdgcn_s_setprio(0);
                STAMP(1)
                lds_barrier();   // every wave is done reading the previous image
                STAMP(2)
            }
            if (!last && !(TRAIN && fused_lean(NT))) FUSED_LOAD_BIAS(l + 1)   // lands while this epilogue computes its sines
            // epilogue: stash z, h = sin(om z) (+ c = om cos(om z) on the last sine layer)
#pragma unroll
            for (int t = 0; t < K::MTW; ++t) {
                const int mt = wm + K::WM * t;
                if (K::EXACT || mt < NT) {
                    // the accumulator IS the phase om z in revolutions (the weights carry om / 2 pi: brief_layout.h); its fraction
                    // is what sin and cos are taken of, here and (TRAIN) again from the stash by the dgrad chain (cos) and by
                    // k_wgrad (sin).  One VALU instruction per element where the exact two-term reduction took five.
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] = __builtin_amdgcn_fractf(acc[t][r]);
                    if (TRAIN && !last) {
                        const __amdgpu_buffer_rsrc_t rz =
                            __builtin_amdgcn_make_buffer_rsrc((void *)(a.Z + (int64_t)l * K::FP * npad), 0, stash_bytes, 0x00020000);
                        // stash planes are tile-blocked, [32-sample tile][FP rows][32 samples]: a tile's block is one contiguous
                        // 4 * 32 * FP bytes (rows 128 B apart), for this kernel's stores and reloads and for k_wgrad's panels
                        const int voff = (int)(n0 * (K::FP * 4)) + ln * 4 + hi * 4 * 128;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            bstore1(acc[t][r], rz, voff, (32 * mt + (r & 3) + 8 * (r >> 2)) * 128);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) hreg[t][r] = BRIEF_SIN_REV(acc[t][r]);
                }
            }
            if (TRAIN && last) {
                // a real (scalar) branch: if-converted, the 32 v_cos of every earlier layer were computed and thrown away
                asm volatile("" ::: "memory");
#pragma unroll
                for (int t = 0; t < K::MTW; ++t) {
                    if (K::EXACT || wm + K::WM * t < NT) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) creg[t][r] = BRIEF_COS_REV(acc[t][r]);      // its om rides on g (gom below)
                    }
                }
            }
            write_image<NT, !TRAIN>(Xs, hreg, wm, lane);
            STAMP(3)
            lds_barrier();
            STAMP(4)
        }
        // ---- head (every wave evaluates it for its sample tile; F MACs per sample)
        float zo[4], yh[4], g[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            zo[c] = 0.f; yh[c] = 0.f; g[c] = 0.f;
            if (c < cout) {
                float p = 0.f;
                const float *wrow = HW + c * K::FP;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 hv = Xs[(kt * 4 + q) * 64 + lane];
                        const float4 wv = *reinterpret_cast<const float4 *>(wrow + 32 * kt + 8 * q + 4 * hi);
                        p = __fmaf_rn(wv.x, hv.x, p); p = __fmaf_rn(wv.y, hv.y, p);
                        p = __fmaf_rn(wv.z, hv.z, p); p = __fmaf_rn(wv.w, hv.w, p);
                    }
                }
                p += __shfl_xor(p, 32);
                zo[c] = p + HW[4 * K::FP + c];
                yh[c] = d.output_act ? brief_fast_sinf(d.w0_hidden * zo[c]) : zo[c];
            }
        }
        if (!TRAIN) {
            if (wm == 0 && hi == 0 && valid) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c >= cout) break;
                    if (a.out_kind == BRIEF_OUT_F32) {
                        reinterpret_cast<float *>(a.out)[n * cout + c] = yh[c];
                    } else {
                        // utils/io.py:136-147: separate roundings, truncating cast
                        float t = __fsub_rn(yh[c], a.scale_min);
                        t = __fdiv_rn(t, a.den);
                        t = fminf(fmaxf(t, 0.f), 1.f);
                        const float u = __fadd_rn(__fmul_rn(t, a.span), a.vmin);
                        if (a.out_kind == BRIEF_OUT_U16) reinterpret_cast<uint16_t *>(a.out)[n * cout + c] = (uint16_t)(int)u;
                        else reinterpret_cast<uint8_t *>(a.out)[n * cout + c] = (uint8_t)(int)u;
                    }
                }
            }
            lds_barrier();   // image is re-used by the next tile
            continue;
        }
        // ---- loss and dloss/dyhat (main.py:176-191)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < cout && valid) {
                float we = wv4[c];
                if (a.thr != 0.f && yh[c] <= a.thr) we = 1.0f;
                const float df = yh[c] - yv[c];
                float li, gi;
                if (a.loss_kind == BRIEF_LOSS_L2) { li = df * df; gi = 2.0f * df; }
                else if (a.loss_kind == BRIEF_LOSS_SMOOTHL1) {
                    const float ad = fabsf(df);
                    if (ad < a.beta) { li = 0.5f * df * df / a.beta; gi = df / a.beta; }
                    else { li = ad - 0.5f * a.beta; gi = df < 0.f ? -1.0f : 1.0f; }
                } else { li = 0.f; gi = 0.f; }
                if (wm == 0 && hi == 0) lsum += li * we;
                g[c] = a.loss_kind == BRIEF_LOSS_EXTERNAL ? yv[c] : gi * we * a.inv_count;      // external: targets ARE dL/dyhat
                if (d.output_act) g[c] *= d.w0_hidden * brief_fast_cosf(d.w0_hidden * zo[c]);
                if (a.yhat_out && wm == 0 && hi == 0) a.yhat_out[n * cout + c] = yh[c];
            }
        }
        STAMP(5)
        // ---- head gradients: transpose own h tiles through LDS (scratch aliases the image: every
        //      wave must be past its head reads first), lane <-> local feature
        lds_barrier();
#pragma unroll
        for (int t = 0; t < K::MTW; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Tw[(32 * t + ROWMAP(r, hi)) * 33 + ln] = hreg[t][r];
        }
        if (hi == 0) *reinterpret_cast<float4 *>(Gw + ln * 4) = make_float4(g[0], g[1], g[2], g[3]);
        else *reinterpret_cast<float4 *>(Gw + 128 + ln * 4) = make_float4(x0, x1, x2, 1.0f);
        lds_barrier();
#pragma unroll
        for (int rs = 0; rs < K::RS; ++rs) {
            float4 sW = make_float4(0.f, 0.f, 0.f, 0.f), sb = make_float4(0.f, 0.f, 0.f, 0.f);
            const int row = lane + 64 * rs;
            const float *trow = Tw + (row < K::TROWS ? row : 0) * 33;
            if (cout == 1) {
                // one channel (every volume BRIEF compresses but RGB images): a quarter of the multiply-adds
#pragma unroll 8
                for (int s = 0; s < 32; ++s) {
                    const float gv = Gw[s * 4];
                    sW.x = __fmaf_rn(trow[s], gv, sW.x);
                    sb.x += gv;
                }
            } else {
#pragma unroll 4
                for (int s = 0; s < 32; ++s) {
                    const float hv = trow[s];
                    const float4 gv = *reinterpret_cast<const float4 *>(Gw + s * 4);
                    sW.x = __fmaf_rn(hv, gv.x, sW.x); sW.y = __fmaf_rn(hv, gv.y, sW.y);
                    sW.z = __fmaf_rn(hv, gv.z, sW.z); sW.w = __fmaf_rn(hv, gv.w, sW.w);
                    sb.x += gv.x; sb.y += gv.y; sb.z += gv.z; sb.w += gv.w;
                }
            }
            if (row < K::TROWS) { accWh[rs][0] += sW.x; accWh[rs][1] += sW.y; accWh[rs][2] += sW.z; accWh[rs][3] += sW.w; }
            if (rs == 0) { accbh[0] += sb.x; accbh[1] += sb.y; accbh[2] += sb.z; accbh[3] += sb.w; }
        }
        // ---- delta of the last sine layer: om cos(phase) * (Wh^T g); the om rides on g (4 multiplies per sample, not 16 MTW)
        f32x16 dl[K::MTW];
        float gom[4];
        {
            const float om_top = (L - 2) == 0 ? d.w0_first : d.w0_hidden;
#pragma unroll
            for (int c = 0; c < 4; ++c) gom[c] = om_top * g[c];
        }
#pragma unroll
        for (int t = 0; t < K::MTW; ++t) {
            const int mt = wm + K::WM * t;
#pragma unroll
            for (int r = 0; r < 16; ++r) dl[t][r] = 0.f;
            if (K::EXACT || mt < NT) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (c < cout) {
                            const float4 wv = *reinterpret_cast<const float4 *>(HW + c * K::FP + 32 * mt + 8 * q + 4 * hi);
                            sacc.x = __fmaf_rn(wv.x, gom[c], sacc.x); sacc.y = __fmaf_rn(wv.y, gom[c], sacc.y);
                            sacc.z = __fmaf_rn(wv.z, gom[c], sacc.z); sacc.w = __fmaf_rn(wv.w, gom[c], sacc.w);
                        }
                    }
                    dl[t][4 * q] = creg[t][4 * q] * sacc.x; dl[t][4 * q + 1] = creg[t][4 * q + 1] * sacc.y;
                    dl[t][4 * q + 2] = creg[t][4 * q + 2] * sacc.z; dl[t][4 * q + 3] = creg[t][4 * q + 3] * sacc.w;
                }
            }
        }
        STAMP(6)
        // ---- dgrad chain: layers L-2 .. 1
        for (int l = L - 2; l >= 1; --l) {
            // stash delta_l for the weight-gradient GEMM and publish it as the B image
            const __amdgpu_buffer_rsrc_t rd =
                __builtin_amdgcn_make_buffer_rsrc((void *)(a.D + (int64_t)(l - 1) * K::FP * npad), 0, stash_bytes, 0x00020000);
            const int voff_s = (int)(n0 * (K::FP * 4)) + ln * 4 + hi * 4 * 128;
#pragma unroll
            for (int t = 0; t < K::MTW; ++t) {
                const int mt = wm + K::WM * t;
                if (K::EXACT || mt < NT) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        bstore1(dl[t][r], rd, voff_s, (32 * mt + (r & 3) + 8 * (r >> 2)) * 128);
                }
            }
            // z_{l-1} comes back from the stash while the chain runs (it is only needed after it)
            const __amdgpu_buffer_rsrc_t rzp =
                __builtin_amdgcn_make_buffer_rsrc((void *)(a.Z + (int64_t)(l - 1) * K::FP * npad), 0, stash_bytes, 0x00020000);
            float zr[K::MTW][16];
            constexpr bool ZPRE = !fused_lean(NT);      // lean variant: the phases are fetched after the chain (two other waves cover the latency)
#define FUSED_LOAD_Z()                                                                                  \
    _Pragma("unroll") for (int t = 0; t < K::MTW; ++t) {                                                \
        const int mt = wm + K::WM * t;                                                                  \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                \
            zr[t][r] = 0.f;                                                                             \
            if (K::EXACT || mt < NT) zr[t][r] = bload1s(rzp, voff_s, (32 * mt + (r & 3) + 8 * (r >> 2)) * 128); \
        }                                                                                               \
    }
            if (ZPRE) FUSED_LOAD_Z()
            STAMP(7)
            lds_barrier();   // transpose scratch / previous chain finished with the image region
            write_image<NT, !TRAIN>(Xs, dl, wm, lane);
            lds_barrier();
            STAMP(8)
#pragma unroll
            for (int t = 0; t < K::MTW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            __builtin_amdgcn_s_setprio(3);
            chain<NT, !TRAIN>(acc, rs_pk, (int)((brief_pk_hidden(d, l) + K::FP * K::FP) * 4), Xs, wm, lane, kit);
            __builtin_amdgcn_s_setprio(0);
            STAMP(9)
            if (!ZPRE) FUSED_LOAD_Z()
#undef FUSED_LOAD_Z
            // delta_{l-1} = acc * cos(phase_{l-1}): the chain ran on w0_{l-1} W_l^T, the stash holds the phase in revolutions
#pragma unroll
            for (int t = 0; t < K::MTW; ++t) {
                const int mt = wm + K::WM * t;
                if (K::EXACT || mt < NT) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        dl[t][r] = acc[t][r] * BRIEF_COS_REV(zr[t][r]);
                }
            }
        }
        STAMP(7)
        // ---- first-layer gradients from delta_0 (lane <-> local feature)
        lds_barrier();
#pragma unroll
        for (int t = 0; t < K::MTW; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Tw[(32 * t + ROWMAP(r, hi)) * 33 + ln] = dl[t][r];
        }
        lds_barrier();
#pragma unroll
        for (int rs = 0; rs < K::RS; ++rs) {
            float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f);
            const int row = lane + 64 * rs;
            const float *trow = Tw + (row < K::TROWS ? row : 0) * 33;
#pragma unroll 4
            for (int s = 0; s < 32; ++s) {
                const float dv = trow[s];
                const float4 xv = *reinterpret_cast<const float4 *>(Gw + 128 + s * 4);
                s0.x = __fmaf_rn(dv, xv.x, s0.x); s0.y = __fmaf_rn(dv, xv.y, s0.y);
                s0.z = __fmaf_rn(dv, xv.z, s0.z); s0.w = __fmaf_rn(dv, xv.w, s0.w);
            }
            if (row < K::TROWS) { acc0[rs][0] += s0.x; acc0[rs][1] += s0.y; acc0[rs][2] += s0.z; acc0[rs][3] += s0.w; }
        }
        lds_barrier();
        STAMP(6)
    }
#undef FUSED_LOAD_BIAS
    if (TRAIN) {
        float *rec = a.rec + ((int64_t)blockIdx.x * 4 + wave) * BRIEF_REC_FLOATS;
#pragma unroll
        for (int rs = 0; rs < K::RS; ++rs) {
            const int row = lane + 64 * rs;
            if (row < 128) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    rec[row * 4 + c] = acc0[rs][c];                        // dW0[f_local][x0,x1,x2,bias]
                    rec[BRIEF_REC_DWH + c * 128 + row] = accWh[rs][c];     // dWh[c][f_local]
                }
            }
        }
        for (int off = 32; off >= 1; off >>= 1) lsum += __shfl_xor(lsum, off);
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) rec[BRIEF_REC_DBH + c] = accbh[c];
            rec[BRIEF_REC_LOSS] = lsum;
#ifdef BRIEF_STAMPS
            for (int i = 0; i < 10; ++i) rec[BRIEF_REC_STAMPS + i] = st_acc[i];
            rec[BRIEF_REC_STAMPS + 10] = (float)(clock64() - st_c0);
            rec[BRIEF_REC_STAMPS + 11] = (float)(wall_clock64() - st_r0);
            rec[BRIEF_REC_STAMPS + 12] = (float)(st_r0 & 0xFFFFFF);      // absolute start (100 MHz ticks, low 24 bits) and the XCC the workgroup ran on
            rec[BRIEF_REC_STAMPS + 13] = (float)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15);      // HW_REG_XCC_ID[3:0]
            rec[BRIEF_REC_STAMPS + 14] = (float)(__builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11)) & 0xFFFF);  // HW_REG_HW_ID[15:0]: wave, simd, pipe, cu[11:8], sh[12], se[15:13]
#endif
        }
    }
}

// ---------------------------------------------------------------------------------------------
// BRIEF_PREC_BF16X3 train step and inference, 64-sample tiles: k_fused<8>'s job on split-precision MFMAs, every A (weight) fragment against TWO B
// fragments (two 32-sample halves of the tile).  A 32-sample tile pulls 256 KB of hi | lo fragments per layer through the CU's
// 64 B/clk vector-memory path — 4 096 cycles per tile-layer against 3 072 cycles of MFMA issue, the chains of the 32-sample
// kernel ran at 7.7 k cycles per layer; here a wave owns 2 feature tiles x 2 sample halves (64 accumulator registers) and the
// same bytes feed twice the MFMAs (a 32-sample form of this kernel lived inside k_fused until the end of round 3: git history,
// profiles/r03_bf16x3.md).  fp16 halves forward, bf16 halves
// backward, f32 stashes in the same tile-blocked planes, same records for k_reduce); only the order in which a workgroup's
// samples enter its skinny-gradient sums differs.  LDS: two hi | lo images (64 KB) + G + head weights + head partials = 76 KB,
// two workgroups per CU.
struct X3TLds {
    static constexpr int IMG_FLOATS = 2 * 4 * 64 * 33;          // two sample halves x (hi | lo) images = 64 KB, aliased by the transposed scratch [2 halves][4 waves][64 rows][33] = 66 KB
    static constexpr int G_OFF = IMG_FLOATS;
    static constexpr int HW_OFF = G_OFF + 4 * 256;
    static constexpr int PART_OFF = HW_OFF + 4 * 256 + 4;       // head partials [2 halves][4 waves][32] float4
    static constexpr int TOTAL = PART_OFF + 2 * 4 * 32 * 4;
};

template <bool F16, bool BPRE>      // BPRE: B fragments a step ahead (16 more registers: the inference kernel has them)
__device__ __forceinline__ void x3_chain2(f32x16 (&acc)[2][2], const X3Pre<8> &pre, __amdgpu_buffer_rsrc_t rs, int soff_layer, int lo_bytes,
                                          const uint4 *X16, int wm, int lane, int kit, bool two)
{
    constexpr int NT = 8, NIT = 16, PD = BRIEF_X3_PD, HALF = 2 * NT * 2 * 64;      // uint4 per sample half (hi + lo)
    using K = KCfg<8>;
    const int voff = lane * 16;
    int soff_w = soff_layer + wm * (NT * 2 * 1024);
    asm volatile("" : "+s"(soff_w));
    int soff_lo = lo_bytes;
    asm volatile("" : "+s"(soff_lo));
    uint4 ahi[NIT][2], alo[NIT][2];
    uint4 bhi[NIT][2], blo[NIT][2];
#pragma unroll
    for (int it = 0; it < PD; ++it)
#pragma unroll
        for (int t = 0; t < 2; ++t) { ahi[it][t] = pre.hi[it][t]; alo[it][t] = pre.lo[it][t]; }
#pragma unroll
    for (int h = 0; h < 2; ++h) { bhi[0][h] = X16[h * HALF + lane]; blo[0][h] = X16[h * HALF + NT * 2 * 64 + lane]; }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        if (it + PD < NIT) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const u32x4 vh = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff_w + (K::WM * t * NT * 2 + it + PD) * 1024, 0);
                const u32x4 vl = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff_w + soff_lo + (K::WM * t * NT * 2 + it + PD) * 1024, 0);
                ahi[it + PD][t] = make_uint4(vh.x, vh.y, vh.z, vh.w);
                alo[it + PD][t] = make_uint4(vl.x, vl.y, vl.z, vl.w);
            }
        }
        // (B fragments are read in the step that uses them: a step ahead they cost 16 more registers, and with those the persistent
        //  per-tile state no longer fits beside the chain — 25 spilled dwords, reloaded from scratch in every head / gradient phase)
        if (BPRE ? it + 1 < NIT : it > 0) {
            constexpr int ahead = BPRE ? 1 : 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bhi[it + ahead][h] = X16[h * HALF + (it + ahead) * 64 + lane];
                blo[it + ahead][h] = X16[h * HALF + NT * 2 * 64 + (it + ahead) * 64 + lane];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (it < kit) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                X3Frag fah, fal;
                fah.u = ahi[it][t]; fal.u = alo[it][t];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h == 1 && !two) continue;      // a single half-tile at the end of a workgroup's run (wave-uniform; no loads inside)
                    X3Frag fbh, fbl;
                    fbh.u = bhi[it][h]; fbl.u = blo[it][h];
                    if (F16) {
                        acc[h][t] = MFMA_X3F(fah.f, fbh.f, acc[h][t]);
                        acc[h][t] = MFMA_X3F(fah.f, fbl.f, acc[h][t]);
                        acc[h][t] = MFMA_X3F(fal.f, fbh.f, acc[h][t]);
                    } else {
                        acc[h][t] = MFMA_X3(fah.v, fbh.v, acc[h][t]);
                        acc[h][t] = MFMA_X3(fah.v, fbl.v, acc[h][t]);
                        acc[h][t] = MFMA_X3(fal.v, fbh.v, acc[h][t]);
                    }
                }
            }
        }
    }
}

template <bool TRAIN>      // false: inference (forward / decode_grid) — the forward half of the same walk, output written by the head
__global__ __launch_bounds__(256, 2) void k_fused_x3(const FusedArgs a)
{
#ifdef BRIEF_STAMPS
    float st_acc[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    long long st_last = clock64();
    const long long st_c0 = st_last, st_r0 = wall_clock64();
#endif
    constexpr int NT = 8, FP = 256, HALF = 2 * NT * 2 * 64;
    using K = KCfg<8>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint4 *X16 = reinterpret_cast<uint4 *>(smem);
    float *T = smem;                       // aliases the images
    float *G = smem + X3TLds::G_OFF;
    float *HW = smem + X3TLds::HW_OFF;     // Whp[4][FP], bhp[4]
    float4 *PART = reinterpret_cast<float4 *>(smem + X3TLds::PART_OFF);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, ln = lane & 31;
    const int wm = wave;
    const brief_siren_desc &d = a.d;
    const int L = d.layers, cin = d.cin, cout = d.cout;
    const int kit16 = (d.features + 15) >> 4;
    const int64_t npad = a.npad;
    const float *pk = a.pk;
    float *Tw = T + wave * 2 * (K::TROWS * 33);
    float *Gw = G + wave * 256;
    const float4 *W0p = reinterpret_cast<const float4 *>(pk + brief_pk_w0(d));
    const __amdgpu_buffer_rsrc_t rs_x3 =
        __builtin_amdgcn_make_buffer_rsrc((void *)(a.pk + brief_pk16_off(a.d, 1)), 0, (int)(2 * brief_pk16_region(a.d) * 4), 0x00020000);
    const int x3_lo_bytes = (int)(brief_pk16_region(a.d) * 4);
    const int stash_bytes = (a.diag & 1) ? 0 : (int)((int64_t)FP * npad * 4);
    {
        const float *headp = pk + brief_pk_head(d);
        for (int e = threadIdx.x; e < 4 * FP + 4; e += 256) HW[e] = headp[e];
    }
    lds_barrier();
    float acc0[4] = {0.f, 0.f, 0.f, 0.f}, accWh[4] = {0.f, 0.f, 0.f, 0.f}, accbh[4] = {0.f, 0.f, 0.f, 0.f};
    float lsum = 0.f;
    // Work is dealt in 32-sample half-tiles walked two at a time.  With whole 64-sample tiles, 100 000 samples are 1 563 tiles for 512
    // resident workgroups — 3.05 rounds, a fourth round at 5 % occupancy.  Instead: R = Hn / (2 G) full rounds of pairs, strided
    // (round r: workgroup w takes half-tiles 2 (r G + w), +1 — the grid writes ONE moving window of the stash planes at a time;
    // contiguous runs per workgroup, 512 scattered write streams, cost 20 k cycles per wave in the delta stores and 12 us in the
    // k_wgrad that follows), then the Hr < 2 G left-over half-tiles as one more pair for the first Hr - G workgroups (if Hr > G)
    // and a single (second half skipped) for the others that get one.
    const int64_t Hn = (a.n + 31) >> 5;
    const int64_t G_ = gridDim.x, w_ = blockIdx.x;
    const int64_t R_ = Hn / (2 * G_), Hr = Hn - 2 * G_ * R_;
    const int64_t npair_x = Hr > G_ ? Hr - G_ : 0;                          // workgroups whose extra is a pair
    const int64_t nsing_x = Hr > G_ ? G_ - npair_x : Hr;                     // ... a single (the workgroups after those)
    const int extra = w_ < npair_x ? 2 : (w_ < npair_x + nsing_x ? 1 : 0);
    const int64_t ht_x = 2 * G_ * R_ + (w_ < npair_x ? 2 * w_ : 2 * npair_x + (w_ - npair_x));
    for (int64_t it_ = 0; it_ < R_ + (extra ? 1 : 0); ++it_) {
        const int64_t ht = it_ < R_ ? 2 * (it_ * G_ + w_) : ht_x;
        const bool two = it_ < R_ || extra == 2;      // workgroup-uniform: the second half exists
        // ---- sample inputs of both halves (kernarg scalars re-read per tile, as in k_fused)
        typedef const __attribute__((address_space(4))) FusedArgs *kargs_f;
        kargs_f ap = (kargs_f)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ap));
        const int64_t *k_idx = ap->idx;
        const float *k_tg = ap->targets, *k_wt = ap->weights, *k_co = ap->coords;
        const uint64_t k_pop = ap->rng_pop, k_seed = ap->rng_seed, k_step = ap->rng_step;
        const int64_t k_off = ap->offset;
        GridArgs kg;
        kg.ndim = ap->grid.ndim; kg.lo = ap->grid.lo; kg.hi = ap->grid.hi; kg.fast = ap->grid.fast;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) { kg.dims[ax] = ap->grid.dims[ax]; kg.step[ax] = ap->grid.step[ax]; kg.magic[ax] = ap->grid.magic[ax]; }
        float x0[2], x1[2], x2[2];
        int64_t jidx[2];      // targets and loss weights are fetched at the loss (two workgroups per CU cover the latency; 16 registers less across the forward pass)
        bool valid[2];
        f32x16 acc[2][2];      // ONE 64-register array per wave: accumulators -> phases -> cos of the last layer -> deltas (in place throughout)
        // The 64 lanes draw the pair's 64 sample indices and coordinates ONCE (lane = sample: lanes 0..31 the first half, 32..63 the
        // second) and then hand every lane both halves' values for its column ln through the cross-lane network; computed per half as
        // in k_fused, the Philox draw, the 64-bit modulo and the grid coordinates ran twice (6 % of the kernel for this phase).
        {
            const int64_t n64 = ht * 32 + lane;
            const bool v64 = (lane < 32 || two) && n64 < a.n;
            int64_t j = 0;
            if (v64) j = k_idx ? k_idx[n64] : (k_pop ? philox_index(n64, k_pop, k_seed, k_step) : n64 + k_off);
            float c0 = 0.f, c1 = 0.f, c2 = 0.f;
            if (v64) {
                if (k_co) {
                    c0 = k_co[j * cin];
                    c1 = k_co[j * cin + 1];
                    if (cin == 3) c2 = k_co[j * cin + 2];
                } else {
                    grid_coords(kg, cin, j, c0, c1, c2);
                }
            }
            const int jl = (int)(uint32_t)(uint64_t)j, jh = (int)(uint32_t)((uint64_t)j >> 32);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int src = 32 * h + ln;
                x0[h] = __shfl(c0, src); x1[h] = __shfl(c1, src); x2[h] = __shfl(c2, src);
                jidx[h] = (int64_t)(((uint64_t)(uint32_t)__shfl(jh, src) << 32) | (uint32_t)__shfl(jl, src));
                valid[h] = (h == 0 || two) && (ht + h) * 32 + ln < a.n;
            }
        }
        X3Pre<8> x3pre;
        float4 bnext[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) bnext[t][q] = make_float4(0.f, 0.f, 0.f, 0.f);
#define X3T_LOAD_BIAS(layer)                                                                            \
    {                                                                                                   \
        const float *bp_ = pk + brief_pk_hidden(d, (layer)) + 2 * FP * FP;                              \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                 \
            const int mt = wm + 4 * t;                                                                  \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                                               \
                bnext[t][q] = *reinterpret_cast<const float4 *>(bp_ + 32 * mt + 8 * q + 4 * hi);        \
        }                                                                                               \
    }
        // ---- layer 0 (exact f32): z0 = W0 x + b0 as two K=2 MFMAs per feature tile and half
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float b0 = hi ? x1[h] : x0[h];
            const float b1 = hi ? 1.0f : x2[h];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int mt = wm + 4 * t;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[h][t][r] = 0.f;
                const float4 w = W0p[32 * mt + ln];
                acc[h][t] = MFMA(hi ? w.y : w.x, b0, acc[h][t]);
                acc[h][t] = MFMA(hi ? w.w : w.z, b1, acc[h][t]);
            }
        }
        STAMP(0)
        // ---- sine layers 0 .. L-2
        for (int l = 0; l <= L - 2; ++l) {
            const bool last = (l == L - 2);
            if (l > 0) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc[h][t][4 * q] = BRIEF_X3_FWD_SCALE * bnext[t][q].x; acc[h][t][4 * q + 1] = BRIEF_X3_FWD_SCALE * bnext[t][q].y;
                            acc[h][t][4 * q + 2] = BRIEF_X3_FWD_SCALE * bnext[t][q].z; acc[h][t][4 * q + 3] = BRIEF_X3_FWD_SCALE * bnext[t][q].w;
                        }
                __builtin_amdgcn_s_setprio(3);
                x3_chain2<true, !TRAIN>(acc, x3pre, rs_x3, (l - 1) * FP * FP * 4, x3_lo_bytes, X16, wm, lane, kit16, two);
                __builtin_amdgcn_s_setprio(0);
                STAMP(1)
                lds_barrier();   // every wave is done reading the previous images
                STAMP(2)
            }
            if (!last) {
                x3_preload<8>(x3pre, rs_x3, l * FP * FP * 4, x3_lo_bytes, wm, lane);      // ahead of this epilogue's stash stores
                X3T_LOAD_BIAS(l + 1)
            }
            float pph[2][4];      // (last layer) head partials of both halves
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int c = 0; c < 4; ++c) pph[h][c] = 0.f;
                if (h == 1 && !two) continue;
                const int voff = (int)(((ht + h) * 32) * (FP * 4)) + ln * 4 + hi * 4 * 128;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int mt = wm + 4 * t;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[h][t][r] = __builtin_amdgcn_fractf(l > 0 ? acc[h][t][r] * BRIEF_X3_FWD_UNSCALE : acc[h][t][r]);
                    f32x16 hv;
                    if (!last) {
                        if (TRAIN) {
                        const __amdgpu_buffer_rsrc_t rz =
                            __builtin_amdgcn_make_buffer_rsrc((void *)(a.Z + (int64_t)l * FP * npad), 0, stash_bytes, 0x00020000);
#pragma unroll
                        for (int r = 0; r < 16; ++r) bstore1s(acc[h][t][r], rz, voff, (32 * mt + (r & 3) + 8 * (r >> 2)) * 128);
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) hv[r] = BRIEF_SIN_REV(acc[h][t][r]);
                        x3_write_tile<true>(X16 + h * HALF, hv, mt, lane);
                    } else {
                        // last sine layer: h feeds the head (from registers) and goes to this wave's transposed scratch for the head
                        // gradients (the images are dead: every wave is past the last chain's barrier); cos(phase) replaces the phase
#pragma unroll
                        for (int r = 0; r < 16; ++r) { hv[r] = BRIEF_SIN_REV(acc[h][t][r]); if (TRAIN) acc[h][t][r] = BRIEF_COS_REV(acc[h][t][r]); }
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (c < cout) {
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const float4 wv = *reinterpret_cast<const float4 *>(HW + c * FP + 32 * mt + 8 * q + 4 * hi);
                                    pph[h][c] = __fmaf_rn(wv.x, hv[4 * q], pph[h][c]); pph[h][c] = __fmaf_rn(wv.y, hv[4 * q + 1], pph[h][c]);
                                    pph[h][c] = __fmaf_rn(wv.z, hv[4 * q + 2], pph[h][c]); pph[h][c] = __fmaf_rn(wv.w, hv[4 * q + 3], pph[h][c]);
                                }
                            }
                        }
                        if (TRAIN) {
                            float *Th = T + (wave * 2 + h) * (K::TROWS * 33);
#pragma unroll
                            for (int r = 0; r < 16; ++r) Th[(32 * t + ROWMAP(r, hi)) * 33 + ln] = hv[r];
                        }
                    }
                }
                if (last) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) pph[h][c] += __shfl_xor(pph[h][c], 32);
                    if (hi == 0) PART[(h * 4 + wave) * 32 + ln] = make_float4(pph[h][0], pph[h][1], pph[h][2], pph[h][3]);
                }
            }
            STAMP(3)
            lds_barrier();      // images (or, last layer, the head partials) are published
            STAMP(4)
        }
        // ---- head: the waves' partial dot products (taken from the exact f32 activations above), summed in wave order
        float zo[2][4], yh[2][4], g[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 tot = PART[(h * 4) * 32 + ln];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 v = PART[(h * 4 + w) * 32 + ln];
                tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w;
            }
            const float tt[4] = {tot.x, tot.y, tot.z, tot.w};
            const int64_t n = (ht + h) * 32 + ln;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                zo[h][c] = 0.f; yh[h][c] = 0.f; g[h][c] = 0.f;
                if (c < cout) {
                    zo[h][c] = tt[c] + HW[4 * FP + c];
                    yh[h][c] = d.output_act ? brief_fast_sinf(d.w0_hidden * zo[h][c]) : zo[h][c];
                }
                if (!TRAIN) {
                    if (c < cout && valid[h] && wm == 0 && hi == 0) {
                        if (a.out_kind == BRIEF_OUT_F32) {
                            reinterpret_cast<float *>(a.out)[n * cout + c] = yh[h][c];
                        } else {
                            // utils/io.py:136-147: separate roundings, truncating cast
                            float t_ = __fsub_rn(yh[h][c], a.scale_min);
                            t_ = __fdiv_rn(t_, a.den);
                            t_ = fminf(fmaxf(t_, 0.f), 1.f);
                            const float u_ = __fadd_rn(__fmul_rn(t_, a.span), a.vmin);
                            if (a.out_kind == BRIEF_OUT_U16) reinterpret_cast<uint16_t *>(a.out)[n * cout + c] = (uint16_t)(int)u_;
                            else reinterpret_cast<uint8_t *>(a.out)[n * cout + c] = (uint8_t)(int)u_;
                        }
                    }
                    continue;
                }
                // ---- loss and dloss/dyhat (main.py:176-191)
                if (c < cout && valid[h]) {
                    const float yvc = k_tg[jidx[h] * cout + c];
                    float we = k_wt ? k_wt[jidx[h] * cout + c] : 1.0f;
                    if (a.thr != 0.f && yh[h][c] <= a.thr) we = 1.0f;
                    const float df = yh[h][c] - yvc;
                    float li, gi;
                    if (a.loss_kind == BRIEF_LOSS_L2) { li = df * df; gi = 2.0f * df; }
                    else if (a.loss_kind == BRIEF_LOSS_SMOOTHL1) {
                        const float ad = fabsf(df);
                        if (ad < a.beta) { li = 0.5f * df * df / a.beta; gi = df / a.beta; }
                        else { li = ad - 0.5f * a.beta; gi = df < 0.f ? -1.0f : 1.0f; }
                    } else { li = 0.f; gi = 0.f; }
                    if (wm == 0 && hi == 0) lsum += li * we;
                    g[h][c] = a.loss_kind == BRIEF_LOSS_EXTERNAL ? yvc : gi * we * a.inv_count;      // external: targets ARE dL/dyhat
                    if (d.output_act) g[h][c] *= d.w0_hidden * brief_fast_cosf(d.w0_hidden * zo[h][c]);
                    if (a.yhat_out && wm == 0 && hi == 0) a.yhat_out[n * cout + c] = yh[h][c];
                }
            }
        }
        STAMP(5)
        if (!TRAIN) continue;      // (the next tile's first LDS writes come after its own barriers; the partials were read above)
        // ---- head gradients, one half after the other, from the transposed h this wave parked in its scratch (lane <-> local feature)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && !two) continue;
            // (no barriers here: the transposed scratch and G are this wave's own, LDS operations of one wave execute in order)
            if (hi == 0) *reinterpret_cast<float4 *>(Gw + ln * 4) = make_float4(g[h][0], g[h][1], g[h][2], g[h][3]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {
                float4 sW = make_float4(0.f, 0.f, 0.f, 0.f), sb = make_float4(0.f, 0.f, 0.f, 0.f);
                const float *trow = T + (wave * 2 + h) * (K::TROWS * 33) + lane * 33;
                if (cout == 1) {
#pragma unroll 8
                    for (int s = 0; s < 32; ++s) {
                        const float gv = Gw[s * 4];
                        sW.x = __fmaf_rn(trow[s], gv, sW.x);
                        sb.x += gv;
                    }
                } else {
#pragma unroll 4
                    for (int s = 0; s < 32; ++s) {
                        const float hv = trow[s];
                        const float4 gv = *reinterpret_cast<const float4 *>(Gw + s * 4);
                        sW.x = __fmaf_rn(hv, gv.x, sW.x); sW.y = __fmaf_rn(hv, gv.y, sW.y);
                        sW.z = __fmaf_rn(hv, gv.z, sW.z); sW.w = __fmaf_rn(hv, gv.w, sW.w);
                        sb.x += gv.x; sb.y += gv.y; sb.z += gv.z; sb.w += gv.w;
                    }
                }
                accWh[0] += sW.x; accWh[1] += sW.y; accWh[2] += sW.z; accWh[3] += sW.w;
                accbh[0] += sb.x; accbh[1] += sb.y; accbh[2] += sb.z; accbh[3] += sb.w;
            }
        }
        // ---- delta of the last sine layer: om cos(phase) * (Wh^T g); the om rides on g
        {
            const float om_top = (L - 2) == 0 ? d.w0_first : d.w0_hidden;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h == 1 && !two) continue;
                float gom[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) gom[c] = om_top * g[h][c];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int mt = wm + 4 * t;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (c < cout) {
                                const float4 wv = *reinterpret_cast<const float4 *>(HW + c * FP + 32 * mt + 8 * q + 4 * hi);
                                sacc.x = __fmaf_rn(wv.x, gom[c], sacc.x); sacc.y = __fmaf_rn(wv.y, gom[c], sacc.y);
                                sacc.z = __fmaf_rn(wv.z, gom[c], sacc.z); sacc.w = __fmaf_rn(wv.w, gom[c], sacc.w);
                            }
                        }
                        acc[h][t][4 * q] *= sacc.x; acc[h][t][4 * q + 1] *= sacc.y;
                        acc[h][t][4 * q + 2] *= sacc.z; acc[h][t][4 * q + 3] *= sacc.w;
                    }
                }
            }
        }
        STAMP(6)
        // ---- dgrad chain: layers L-2 .. 1
        for (int l = L - 2; l >= 1; --l) {
            const __amdgpu_buffer_rsrc_t rd =
                __builtin_amdgcn_make_buffer_rsrc((void *)(a.D + (int64_t)(l - 1) * FP * npad), 0, stash_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rzp =
                __builtin_amdgcn_make_buffer_rsrc((void *)(a.Z + (int64_t)(l - 1) * FP * npad), 0, stash_bytes, 0x00020000);
            x3_preload<8>(x3pre, rs_x3, (l - 1) * FP * FP * 4 + FP * FP * 2, x3_lo_bytes, wm, lane);      // ahead of the delta stores
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h == 1 && !two) continue;
                const int voff_s = (int)(((ht + h) * 32) * (FP * 4)) + ln * 4 + hi * 4 * 128;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int mt = wm + 4 * t;
#pragma unroll
                    for (int r = 0; r < 16; ++r) bstore1s(acc[h][t][r], rd, voff_s, (32 * mt + (r & 3) + 8 * (r >> 2)) * 128);
                }
            }
            STAMP(7)
            lds_barrier();   // transpose scratch / previous chain finished with the image region
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (h == 0 || two) x3_write_image<8, false>(X16 + h * HALF, acc[h], wm, lane);
            lds_barrier();
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[h][t][r] = 0.f;
            STAMP(8)
            __builtin_amdgcn_s_setprio(3);
            x3_chain2<false, false>(acc, x3pre, rs_x3, (l - 1) * FP * FP * 4 + FP * FP * 2, x3_lo_bytes, X16, wm, lane, kit16, two);
            __builtin_amdgcn_s_setprio(0);
            STAMP(9)
            // delta_{l-1} = acc * cos(phase_{l-1}): the phases come back from the stash after the chain, a half (32 requests) at a time
            // (all 64 at once: 152 B of scratch; a feature tile at a time: four exposed round trips per layer)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h == 1 && !two) continue;
                float zr[2][16];
                const int voff_s = (int)(((ht + h) * 32) * (FP * 4)) + ln * 4 + hi * 4 * 128;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int mt = wm + 4 * t;
#pragma unroll
                    for (int r = 0; r < 16; ++r) zr[t][r] = bload1s(rzp, voff_s, (32 * mt + (r & 3) + 8 * (r >> 2)) * 128);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[h][t][r] *= BRIEF_COS_REV(zr[t][r]);
            }
        }
        STAMP(7)
        // ---- first-layer gradients from delta_0, one half after the other (lane <-> local feature)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && !two) continue;
            if (h == 0) lds_barrier();      // the scratch aliases the images: every wave must be past the last dgrad chain
            float *Tf = Tw + h * (K::TROWS * 33);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) Tf[(32 * t + ROWMAP(r, hi)) * 33 + ln] = acc[h][t][r];
            if (hi == 1) *reinterpret_cast<float4 *>(Gw + 128 + ln * 4) = make_float4(x0[h], x1[h], x2[h], 1.0f);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // own-wave scratch: ordering only
            float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f);
            const float *trow = Tf + lane * 33;
#pragma unroll 4
            for (int s = 0; s < 32; ++s) {
                const float dv = trow[s];
                const float4 xv = *reinterpret_cast<const float4 *>(Gw + 128 + s * 4);
                s0.x = __fmaf_rn(dv, xv.x, s0.x); s0.y = __fmaf_rn(dv, xv.y, s0.y);
                s0.z = __fmaf_rn(dv, xv.z, s0.z); s0.w = __fmaf_rn(dv, xv.w, s0.w);
            }
            acc0[0] += s0.x; acc0[1] += s0.y; acc0[2] += s0.z; acc0[3] += s0.w;
        }
        lds_barrier();
        STAMP(6)
    }
#undef X3T_LOAD_BIAS
    if (!TRAIN) return;
    float *rec = a.rec + ((int64_t)blockIdx.x * 4 + wave) * BRIEF_REC_FLOATS;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        rec[lane * 4 + c] = acc0[c];                        // dW0[f_local][x0,x1,x2,bias]
        rec[BRIEF_REC_DWH + c * 128 + lane] = accWh[c];     // dWh[c][f_local]
    }
    for (int off = 32; off >= 1; off >>= 1) lsum += __shfl_xor(lsum, off);
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) rec[BRIEF_REC_DBH + c] = accbh[c];
        rec[BRIEF_REC_LOSS] = lsum;
#ifdef BRIEF_STAMPS
        for (int i = 0; i < 10; ++i) rec[BRIEF_REC_STAMPS + i] = st_acc[i];
        rec[BRIEF_REC_STAMPS + 10] = (float)(clock64() - st_c0);
        rec[BRIEF_REC_STAMPS + 11] = (float)(wall_clock64() - st_r0);
#endif
    }
}

#include "brief_lean.inc"

// ---------------------------------------------------------------------------------------------
// k_small<NT, HB>: the whole train step of a NARROW net (F <= 64: NT = 1, 2; at most HB hidden F x F layers)
// without any HBM stash.  For these widths (every net the reference's own YAMLs produce: F = 22 ... 56) the
// stash traffic of k_fused + k_wgrad, not the matrix pipe, sets the step time (1.8 KB per sample against
// 9 kFLOP), so here everything stays on chip:
//   * the pre-activations z_l of every sine layer stay in registers (16 per layer: a wave owns one 32x32 tile),
//     sin / cos are recomputed from them on the way back;
//   * the weight gradients accumulate in registers across all tiles a workgroup walks: delta_l and h_{l-1}
//     are transposed through LDS ([feature][sample] panels, stride 36 floats, conflict-free ds_read_b128) into
//     A / B operands whose k index is the sample, 16 MFMAs per (32x32 dW tile, 32 samples);
//   * NT = 1: the 4 waves are independent (own samples, own LDS regions, no barriers in the tile loop) and
//     fold their dW at the end;  NT = 2: wave (wm, ws) owns dW tile (wm, ws) over both sample tiles;
//   * one slab per workgroup and layer goes to k_reduce (same slab format as k_wgrad's).
// inputs of sample n for the train kernels: coordinates, targets, loss weights (defaults past the end of the batch)
// (The scalars it needs are re-read from the kernarg segment through an opaque pointer at every call: kept live across a
//  tile loop they get spilled to VGPR lanes and come back one v_readlane — a VALU instruction — at a time.)
typedef const __attribute__((address_space(4))) FusedArgs *kargs_t;
// AP: where the job's FusedArgs live — the kernarg segment (k_small: the kernel's only argument) or a table entry in global memory
// (k_small_group: one entry per co-trained job); k_idx / rng_step: this step's index set and Philox step (per-step values)
template <typename AP>
__device__ __forceinline__ void small_inputs(AP ap, const int64_t *k_idx, uint64_t rng_step, int cin, int cout, int64_t n, float4 &xo, float4 &yo, float4 &wo)
{
    asm volatile("" : "+s"(ap));
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    float yv[4] = {0.f, 0.f, 0.f, 0.f}, wv4[4] = {1.f, 1.f, 1.f, 1.f};
    if (n < ap->n) {
        const float *k_tg = ap->targets, *k_wt = ap->weights, *k_co = ap->coords;
        const uint64_t k_pop = ap->rng_pop;
        const int64_t j = k_idx ? k_idx[n] : (k_pop ? philox_index(n, k_pop, ap->rng_seed, rng_step) : n + ap->offset);
        if (!k_tg) {
            // forward-only launch: no targets
        } else if (cout == 1) {
            yv[0] = k_tg[j];
            if (k_wt) wv4[0] = k_wt[j];
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c < cout) {
                    yv[c] = k_tg[j * cout + c];
                    if (k_wt) wv4[c] = k_wt[j * cout + c];
                }
            }
        }
        if (k_co) {
            x0 = k_co[j * cin];
            x1 = k_co[j * cin + 1];
            if (cin == 3) x2 = k_co[j * cin + 2];
        } else {
            GridArgs kg;
            kg.ndim = ap->grid.ndim; kg.lo = ap->grid.lo; kg.hi = ap->grid.hi; kg.fast = ap->grid.fast;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) { kg.dims[ax] = ap->grid.dims[ax]; kg.step[ax] = ap->grid.step[ax]; kg.magic[ax] = ap->grid.magic[ax]; }
            grid_coords(kg, cin, j, x0, x1, x2);
        }
    }
    xo = make_float4(x0, x1, x2, 0.f);
    yo = make_float4(yv[0], yv[1], yv[2], yv[3]);
    wo = make_float4(wv4[0], wv4[1], wv4[2], wv4[3]);
}

template <int NT>
struct SmallLds {
    using K = KCfg<NT>;
    static constexpr int PANEL = 32 * 36;
    static constexpr int DT_OFF = K::XS_FLOATS;           // delta^T panels, one per wave
    static constexpr int HT_OFF = DT_OFF + 4 * PANEL;     // h^T panels, one per wave
    static constexpr int G_OFF = HT_OFF + 4 * PANEL;
    static constexpr int HW_OFF = G_OFF + 4 * 256;
    static constexpr int TOTAL = HW_OFF + 4 * K::FP + 4;
};
constexpr int small_wpe(int HB) { return HB <= 3 ? 2 : 1; }     // resident workgroups per CU (register budget; three for the one-hidden-layer bucket — 168 VGPRs, 56 B of
                                                                // scratch — measured slower: C1 0.128 against 0.102 ms per step)

// bid / gdim: this workgroup's number among, and the count of, the workgroups that serve THIS job (the whole grid for k_small; a
// contiguous range of it for k_small_group)
template <int NT, int HB, typename AP>
__device__ __forceinline__ void small_body(const FusedArgs &a, AP ap_in, const int64_t *step_idx, uint64_t step_rng, int bid, int gdim)
{
    static_assert(NT == 1 || NT == 2, "narrow nets only");
#ifdef BRIEF_STAMPS
    float st_acc[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    long long st_last = clock64();
#endif
    using K = KCfg<NT>;
    using LD = SmallLds<NT>;
    constexpr int PANEL = LD::PANEL;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4 *X = reinterpret_cast<float4 *>(smem);
    float *DT = smem + LD::DT_OFF, *HT = smem + LD::HT_OFF;
    float *G = smem + LD::G_OFF;
    float *HW = smem + LD::HW_OFF;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, ln = lane & 31;
    const int wm = wave % K::WM, ws = wave / K::WM;
    const brief_siren_desc &d = a.d;
    const int L = d.layers, cin = d.cin, cout = d.cout;
    const int kit = (d.features + 7) >> 3;       // chain steps that hold real (unpadded) features
    const float *pk = a.pk;
    float4 *Xs = X + ws * (NT * 256);
    float *DTw = DT + wave * PANEL, *HTw = HT + wave * PANEL;
    float *Gw = G + wave * 256;
    const float4 *W0p = reinterpret_cast<const float4 *>(pk + brief_pk_w0(d));
    const __amdgpu_buffer_rsrc_t rs_pk =
        __builtin_amdgcn_make_buffer_rsrc((void *)pk, 0, (int)(brief_pk_count(d) * 4), 0x00020000);
    // NT == 1: every LDS region is private to its wave and a wave's LDS operations complete in order
#define TILE_BARRIER() { if (NT > 1) lds_barrier(); else asm volatile("" ::: "memory"); }

    {
        const float *headp = pk + brief_pk_head(d);
        for (int e = threadIdx.x; e < 4 * K::FP + 4; e += 256) HW[e] = headp[e];
    }
    lds_barrier();

    f32x16 dW[HB];
    float dbv[HB];
#pragma unroll
    for (int i = 0; i < HB; ++i) {
        dbv[i] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) dW[i][r] = 0.f;
    }
    // skinny gradients: lane (ln, hi) <-> feature ln of the own tile, samples 16 hi .. 16 hi + 15
    float acc0[4] = {0.f, 0.f, 0.f, 0.f}, accWh[4] = {0.f, 0.f, 0.f, 0.f}, accbh[4] = {0.f, 0.f, 0.f, 0.f};
    float lsum = 0.f;
    const int prow = ln * 36 + 16 * hi;      // this lane's 16 samples of row ln in a panel

    const int64_t wg_samples = 32 * K::WS;
    const int64_t ntiles = (a.n + wg_samples - 1) / wg_samples;
    // (measured and not kept: fetching the next tile's inputs, or the next chain's first A fragments, one
    //  phase ahead costs more in registers -> scratch than the latency it hides)
    for (int64_t tile = bid; tile < ntiles; tile += gdim) {
        const int64_t n0 = (tile * K::WS + ws) * 32;
        const int64_t n = n0 + ln;
        const bool valid = n < a.n;
        float4 in_x, in_y, in_w;     // coords (x0,x1,x2,-) | targets | loss weights
        small_inputs(ap_in, step_idx, step_rng, cin, cout, n, in_x, in_y, in_w);
        const float x0 = in_x.x, x1 = in_x.y, x2 = in_x.z;
        const float yv[4] = {in_y.x, in_y.y, in_y.z, in_y.w}, wv4[4] = {in_w.x, in_w.y, in_w.z, in_w.w};
        f32x16 acc[1], hreg[1];
        f32x16 zst[HB + 1];
        float4 bnext[4];

#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) bnext[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        // ---- layer 0 (two K=2 MFMAs against [x0 x1 | x2 1], bias folded in)
        {
            const float4 w = W0p[32 * wm + ln];
            acc[0] = MFMA(hi ? w.y : w.x, hi ? x1 : x0, acc[0]);
            acc[0] = MFMA(hi ? w.w : w.z, hi ? 1.0f : x2, acc[0]);
        }
        STAMP(0)
        // ---- sine layers 0 .. L-2 (unrolled to the bucket size so that zst[] stays in registers)
#pragma unroll
        for (int l = 0; l <= HB; ++l) {
            if (l <= L - 2) {
                const bool last = (l == L - 2);
                if (l > 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[0][4 * q] = bnext[q].x; acc[0][4 * q + 1] = bnext[q].y;
                        acc[0][4 * q + 2] = bnext[q].z; acc[0][4 * q + 3] = bnext[q].w;
                    }
                    chain<NT>(acc, rs_pk, (int)(brief_pk_hidden(d, l) * 4), Xs, wm, lane, kit);
                    STAMP(1)
                    TILE_BARRIER()
                }
                if (!last) {
                    const float *bp_ = pk + brief_pk_hidden(d, l + 1) + 2 * K::FP * K::FP;
#pragma unroll
                    for (int q = 0; q < 4; ++q) bnext[q] = *reinterpret_cast<const float4 *>(bp_ + 32 * wm + 8 * q + 4 * hi);
                }
                // the accumulator is the phase om z in revolutions (the weights carry om / 2 pi); keep its fraction: the
                // backward pass takes sin and cos of it again
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] = __builtin_amdgcn_fractf(acc[0][r]);
                zst[l] = acc[0];
#pragma unroll
                for (int r = 0; r < 16; ++r) hreg[0][r] = BRIEF_SIN_REV(acc[0][r]);
                write_image<NT>(Xs, hreg, wm, lane);
                TILE_BARRIER()
                STAMP(2)
            }
        }
        // ---- head
        float zo[4], yh[4], g[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            zo[c] = 0.f; yh[c] = 0.f; g[c] = 0.f;
            if (c < cout) {
                float p = 0.f;
                const float *wrow = HW + c * K::FP;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 hv = Xs[(kt * 4 + q) * 64 + lane];
                        const float4 wv = *reinterpret_cast<const float4 *>(wrow + 32 * kt + 8 * q + 4 * hi);
                        p = __fmaf_rn(wv.x, hv.x, p); p = __fmaf_rn(wv.y, hv.y, p);
                        p = __fmaf_rn(wv.z, hv.z, p); p = __fmaf_rn(wv.w, hv.w, p);
                    }
                }
                p += __shfl_xor(p, 32);
                zo[c] = p + HW[4 * K::FP + c];
                yh[c] = d.output_act ? brief_fast_sinf(d.w0_hidden * zo[c]) : zo[c];
            }
        }
        // ---- loss and dloss/dyhat (main.py:176-191)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < cout && valid) {
                float we = wv4[c];
                if (a.thr != 0.f && yh[c] <= a.thr) we = 1.0f;
                const float df = yh[c] - yv[c];
                float li, gi;
                if (a.loss_kind == BRIEF_LOSS_L2) { li = df * df; gi = 2.0f * df; }
                else if (a.loss_kind == BRIEF_LOSS_SMOOTHL1) {
                    const float ad = fabsf(df);
                    if (ad < a.beta) { li = 0.5f * df * df / a.beta; gi = df / a.beta; }
                    else { li = ad - 0.5f * a.beta; gi = df < 0.f ? -1.0f : 1.0f; }
                } else { li = 0.f; gi = 0.f; }
                if (wm == 0 && hi == 0) lsum += li * we;
                g[c] = a.loss_kind == BRIEF_LOSS_EXTERNAL ? yv[c] : gi * we * a.inv_count;      // external: targets ARE dL/dyhat
                if (d.output_act) g[c] *= d.w0_hidden * brief_fast_cosf(d.w0_hidden * zo[c]);
                if (a.yhat_out && wm == 0 && hi == 0) a.yhat_out[n * cout + c] = yh[c];
            }
        }
        // ---- head gradients from the transposed last activation
#pragma unroll
        for (int r = 0; r < 16; ++r) HTw[ROWMAP(r, hi) * 36 + ln] = hreg[0][r];
        if (hi == 0) *reinterpret_cast<float4 *>(Gw + ln * 4) = make_float4(g[0], g[1], g[2], g[3]);
        else *reinterpret_cast<float4 *>(Gw + 128 + ln * 4) = make_float4(x0, x1, x2, 1.0f);
        TILE_BARRIER()      // also: every wave is past its head reads of the image
        if (cout == 1) {
            float sW = 0.f, sb = 0.f;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 hv = *reinterpret_cast<const float4 *>(HTw + prow + 4 * q4);
                const float hvv[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float gv = Gw[(16 * hi + 4 * q4 + jj) * 4];
                    sW = __fmaf_rn(hvv[jj], gv, sW);
                    sb += gv;
                }
            }
            accWh[0] += sW;
            accbh[0] += sb;
        } else {
            float4 sW = make_float4(0.f, 0.f, 0.f, 0.f), sb = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 hv = *reinterpret_cast<const float4 *>(HTw + prow + 4 * q4);
                const float hvv[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float4 gv = *reinterpret_cast<const float4 *>(Gw + (16 * hi + 4 * q4 + jj) * 4);
                    sW.x = __fmaf_rn(hvv[jj], gv.x, sW.x); sW.y = __fmaf_rn(hvv[jj], gv.y, sW.y);
                    sW.z = __fmaf_rn(hvv[jj], gv.z, sW.z); sW.w = __fmaf_rn(hvv[jj], gv.w, sW.w);
                    sb.x += gv.x; sb.y += gv.y; sb.z += gv.z; sb.w += gv.w;
                }
            }
            accWh[0] += sW.x; accWh[1] += sW.y; accWh[2] += sW.z; accWh[3] += sW.w;
            accbh[0] += sb.x; accbh[1] += sb.y; accbh[2] += sb.z; accbh[3] += sb.w;
        }
        STAMP(3)
        // ---- Wh^T g: the last sine layer's delta is c_{L-2} times this (taken inside the unrolled loop below,
        //      where the layer index is a compile-time constant and z comes straight out of its registers)
        f32x16 dl[1];
        float gom[4];      // om of the last sine layer rides on g
        {
            const float om_top = (L - 2) == 0 ? d.w0_first : d.w0_hidden;
#pragma unroll
            for (int c = 0; c < 4; ++c) gom[c] = om_top * g[c];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c < cout) {
                    const float4 wv = *reinterpret_cast<const float4 *>(HW + c * K::FP + 32 * wm + 8 * q + 4 * hi);
                    sacc.x = __fmaf_rn(wv.x, gom[c], sacc.x); sacc.y = __fmaf_rn(wv.y, gom[c], sacc.y);
                    sacc.z = __fmaf_rn(wv.z, gom[c], sacc.z); sacc.w = __fmaf_rn(wv.w, gom[c], sacc.w);
                }
            }
            dl[0][4 * q] = sacc.x; dl[0][4 * q + 1] = sacc.y; dl[0][4 * q + 2] = sacc.z; dl[0][4 * q + 3] = sacc.w;
        }
#define SMALL_TOP_DELTA(layer)                                                                           \
    if ((layer) == L - 2) {                                                                              \
        _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                   \
            dl[0][r] *= BRIEF_COS_REV(zst[(layer)][r]);                                                  \
    }
        // ---- backward through the hidden layers L-2 .. 1: dW_l += delta_l h_{l-1}^T, delta_{l-1} = (W_l^T delta_l) . c_{l-1}
#pragma unroll
        for (int li = HB; li >= 1; --li) {
            SMALL_TOP_DELTA(li)
            if (li <= L - 2) {
                f32x16 cp;      // cos(phase_{li-1}): its om rides on the W^T copy the chain below runs on
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float fr = zst[li - 1][r];
                    HTw[ROWMAP(r, hi) * 36 + ln] = BRIEF_SIN_REV(fr);
                    cp[r] = BRIEF_COS_REV(fr);
                    DTw[ROWMAP(r, hi) * 36 + ln] = dl[0][r];
                }
                write_image<NT>(Xs, dl, wm, lane);
                TILE_BARRIER()
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
                STAMP(4)
                chain<NT>(acc, rs_pk, (int)((brief_pk_hidden(d, li) + K::FP * K::FP) * 4), Xs, wm, lane, kit);
                STAMP(5)
                // weight gradient: k = sample; lane (m, hi) feeds samples 16 hi + s at step s
#pragma unroll
                for (int st = 0; st < K::WS; ++st) {
                    if (NT == 1 && st != ws) continue;       // NT == 1: only the wave's own sample tile
                    const float *pa = (NT == 1 ? DTw : DT + (st * K::WM + wm) * PANEL) + prow;
                    const float *pb = (NT == 1 ? HTw : HT + (st * K::WM + ws) * PANEL) + prow;
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const float4 a4 = *reinterpret_cast<const float4 *>(pa + 4 * q4);
                        const float4 b4 = *reinterpret_cast<const float4 *>(pb + 4 * q4);
                        dW[li - 1] = MFMA(a4.x, b4.x, dW[li - 1]);
                        dW[li - 1] = MFMA(a4.y, b4.y, dW[li - 1]);
                        dW[li - 1] = MFMA(a4.z, b4.z, dW[li - 1]);
                        dW[li - 1] = MFMA(a4.w, b4.w, dW[li - 1]);
                        dbv[li - 1] += (a4.x + a4.y) + (a4.z + a4.w);
                    }
                }
                STAMP(6)
#pragma unroll
                for (int r = 0; r < 16; ++r) dl[0][r] = acc[0][r] * cp[r];
                TILE_BARRIER()      // panels and image are rewritten by the next layer
            }
        }
        SMALL_TOP_DELTA(0)
#undef SMALL_TOP_DELTA
        // ---- first-layer gradients from delta_0
#pragma unroll
        for (int r = 0; r < 16; ++r) DTw[ROWMAP(r, hi) * 36 + ln] = dl[0][r];
        TILE_BARRIER()
        {
            float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 dv = *reinterpret_cast<const float4 *>(DTw + prow + 4 * q4);
                const float dvv[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float4 xv = *reinterpret_cast<const float4 *>(Gw + 128 + (16 * hi + 4 * q4 + jj) * 4);
                    s0.x = __fmaf_rn(dvv[jj], xv.x, s0.x); s0.y = __fmaf_rn(dvv[jj], xv.y, s0.y);
                    s0.z = __fmaf_rn(dvv[jj], xv.z, s0.z); s0.w = __fmaf_rn(dvv[jj], xv.w, s0.w);
                }
            }
            acc0[0] += s0.x; acc0[1] += s0.y; acc0[2] += s0.z; acc0[3] += s0.w;
        }
        TILE_BARRIER()
        STAMP(7)
    }
#undef TILE_BARRIER
    // ---- per-wave record of the skinny gradients (format of k_fused: k_reduce reads both)
    {
        float *rec = a.rec + ((int64_t)bid * 4 + wave) * BRIEF_REC_FLOATS;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float v0 = acc0[c] + __shfl_xor(acc0[c], 32);
            const float vh = accWh[c] + __shfl_xor(accWh[c], 32);
            const float vb = accbh[c] + __shfl_xor(accbh[c], 32);
            if (hi == 0) {
                rec[ln * 4 + c] = v0;
                rec[BRIEF_REC_DWH + c * 128 + ln] = vh;
            }
            if (lane == 0) rec[BRIEF_REC_DBH + c] = vb;
        }
        for (int off = 32; off >= 1; off >>= 1) lsum += __shfl_xor(lsum, off);
        if (lane == 0) rec[BRIEF_REC_LOSS] = lsum;
#ifdef BRIEF_STAMPS
        if (lane == 0) for (int i = 0; i < 10; ++i) rec[BRIEF_REC_STAMPS + i] = st_acc[i];
#endif
    }
    // ---- hidden-layer partials: one slab per (layer, workgroup)
    const int64_t slab_sz = (int64_t)K::FP * K::FP + K::FP;
    if (NT == 2) {
#pragma unroll
        for (int i = 0; i < HB; ++i) {
            if (i < L - 2) {
                float *slab = a.slabs + ((int64_t)i * gdim + bid) * slab_sz;
#pragma unroll
                for (int r = 0; r < 16; ++r) slab[(32 * wm + ROWMAP(r, hi)) * K::FP + 32 * ws + ln] = dW[i][r];
                const float tot = dbv[i] + __shfl_xor(dbv[i], 32);
                if (ws == 0 && hi == 0) slab[K::FP * K::FP + 32 * wm + ln] = tot;
            }
        }
    } else {
        // fold the four waves' partial tiles in wave order (the tile loop's LDS regions are dead now)
        float *red = smem, *redb = smem + 3 * 1024;
        lds_barrier();
#pragma unroll
        for (int i = 0; i < HB; ++i) {
            if (i < L - 2) {
                if (wave > 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(wave - 1) * 1024 + r * 64 + lane] = dW[i][r];
                    redb[(wave - 1) * 64 + lane] = dbv[i];
                }
                lds_barrier();
                if (wave == 0) {
                    f32x16 tsum = dW[i];
                    float bsum = dbv[i];
                    for (int w = 1; w < 4; ++w) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) tsum[r] += red[(w - 1) * 1024 + r * 64 + lane];
                        bsum += redb[(w - 1) * 64 + lane];
                    }
                    float *slab = a.slabs + ((int64_t)i * gdim + bid) * slab_sz;
#pragma unroll
                    for (int r = 0; r < 16; ++r) slab[ROWMAP(r, hi) * K::FP + ln] = tsum[r];
                    const float tot = bsum + __shfl_xor(bsum, 32);
                    if (hi == 0) slab[K::FP * K::FP + ln] = tot;
                }
                lds_barrier();
            }
        }
    }
}

template <int NT, int HB>
__global__ __launch_bounds__(256, small_wpe(HB)) void k_small(const FusedArgs a)
{
    kargs_t ap = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();      // FusedArgs is the kernel's only argument
    small_body<NT, HB>(a, ap, a.idx, a.rng_step, (int)blockIdx.x, (int)gridDim.x);
}

// Co-trained narrow nets in ONE launch (brief_multi_fit; the per-block loop of main.py:547-575 for the blocks one GPU owns): job j is
// served by workgroups [wg_begin[j], wg_begin[j + 1]) — exactly the grid, tile walk, records and slabs its own k_small launch would have,
// so every job's results are bit-identical to a fit on its own.  The static part of a job (pointers, net, grid, loss) sits in a device
// table uploaded once per brief_multi_fit call; what changes from step to step travels in the kernel arguments.
#define BRIEF_GROUP_MAX 64
struct SmallGroupArgs {
    const FusedArgs *table;
    int njobs;
    int wg_begin[BRIEF_GROUP_MAX + 1];
    uint64_t rng_step[BRIEF_GROUP_MAX];
    const int64_t *idx[BRIEF_GROUP_MAX];
};
// dst = *src, with src read through the constant address space (dword by dword: scalar loads, the copy dissolves into SGPRs)
template <typename T>
__device__ __forceinline__ void const_copy(T &dst, const T *src)
{
    static_assert(sizeof(T) % 4 == 0, "dword-sized arguments");
    const __attribute__((address_space(4))) uint32_t *s4 = (const __attribute__((address_space(4))) uint32_t *)src;
    uint32_t *d4 = reinterpret_cast<uint32_t *>(&dst);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); ++i) d4[i] = s4[i];
}
__device__ __forceinline__ int group_job(const int *begin, int njobs, int b)
{
    int j = 0;
    for (int k = 1; k < njobs; ++k) j += (b >= begin[k]);      // begin[] is ascending
    return j;
}
template <int NT, int HB>
__global__ __launch_bounds__(256, small_wpe(HB)) void k_small_group(const SmallGroupArgs g)
{
    const int b = (int)blockIdx.x;
    const int j = group_job(g.wg_begin, g.njobs, b);
    // the table entry is read through the CONSTANT address space (written by an earlier launch, never during this one): scalar loads
    // into SGPRs exactly like kernel arguments.  As plain global memory every use of an argument inside the tile loop was a vector
    // load + v_readfirstlane behind a wait (the kernel stores through pointers it loaded from the same table, so nothing could be
    // hoisted): 197 us per 64^3 job of a 4x35 net in a group of eight against 150 us on its own.
    kargs_t tp = (kargs_t)(g.table + j);
    FusedArgs a;
    const_copy(a, g.table + j);
    small_body<NT, HB>(a, tp, g.idx[j], g.rng_step[j], b - g.wg_begin[j], g.wg_begin[j + 1] - g.wg_begin[j]);
}

// ---------------------------------------------------------------------------------------------
// (the bf16 kernels are included after the optimizer helpers below)
// weight-gradient GEMM: dW_l[fo][fi] = sum_n D_l[fo][n] * sin(2 pi Z_{l-1}[fi][n]),  db_l = sum_n D_l   (Z: revolutions)
struct WgradArgs {
    brief_siren_desc d;
    const float *Z;
    const float *D;
    int64_t npad;
    int nsplit;
    float *slabs;       // [(L-2)][nsplit][FP*FP + FP]
    float *stamps;      // diagnostic build (-DBRIEF_STAMPS) only: [blocks][8 waves][8]
};

// k_wgrad<NT> geometry.  Widths above 8 tiles are cut into NQ x NQ output quadrants of QT x QT tiles, one
// workgroup each (a 512x512 fp32 dW does not fit eight waves' registers); QT plays NT's role inside.
constexpr int wgrad_nq(int NT) { return (NT + 7) / 8; }
constexpr int wgrad_qt(int NT) { return (NT + wgrad_nq(NT) - 1) / wgrad_nq(NT); }      // exact for every compiled width (1 .. 8, 12, 16); above 16 tiles
                                                                                        // (run-time width) the last quadrant row / column may be short
// How the 8 waves of a k_wgrad workgroup cover a QT x QT tile block.  Rectangular form: WMk x WNk wave tiles of TM x TN 32 x 32 tiles each,
// times a WK-way split of every chunk's four k-groups (folded through LDS at the end) — 1, 2, 4, 6 and 8 tiles per side, where it leaves no
// slot empty (6: 2 x 2 waves x 3 x 3 tiles x two k-slices; 8: 2 x 4 waves x 4 x 2).  An odd side has no such grid (7: 64 slots for 49 tiles,
// 5: 36 for 25, 3: 16 for 9), so there the waves take UNEQUAL pieces, sized so that the four SIMDs (which carry waves w and w + 4) end up level:
//   LIST (3 and 5 per side)  wave w owns a run of floor(QT^2 / 8) tiles in row-major order, the last QT^2 % 8 waves one more; no k-split, no fold
//                            (4x96 step 0.392 -> 0.432 of the fp32 peak, 4x160 0.567 -> 0.608, 4x288 0.591 -> 0.649, 4x320 0.659 -> 0.710)
//   HET7 (7 per side)        wave w < 7 owns the first six tiles of tile row w, wave 7 the whole tile column 6: 12, 12, 12, 13 tiles per SIMD
//                            where 16 slots were (4x224 0.647 -> 0.677, 4x448 0.736 -> 0.770, 4x640 0.693 -> 0.739, 4x896 0.733 -> 0.783)
// Each role is a copy of the whole chunk loop (k_wgrad's chunk_loop): a role test inside the loop made the compiler carry two sets of
// accumulators across the merge and spill (measured 20 - 25 % slower than the rectangular form it was meant to beat).  LIST at 4 and 6 per
// side is a wash (+-1 %), at 7 it spills (two fragment reads per tile): BRIEF_WGRAD_LIST picks the sides at compile time for such A/B runs.
#ifndef BRIEF_WGRAD_LIST
#define BRIEF_WGRAD_LIST(QT) ((QT) == 3 || (QT) == 5)
#endif
constexpr int wgrad_wmk(int QT) { return QT >= 2 ? 2 : 1; }
constexpr int wgrad_wnk(int QT) { return QT >= 7 ? 4 : (QT >= 2 ? 2 : 1); }
constexpr bool wgrad_list(int QT) { return BRIEF_WGRAD_LIST(QT); }      // LIST mode (below): every wave owns a run of the QT^2 tiles in row-major order
constexpr int wgrad_wk(int QT) { return QT >= 7 || wgrad_list(QT) ? 1 : (QT >= 2 ? 2 : 4); }
// dynamic LDS in floats: two double-buffered panel pairs (the k-slice fold goes through them one row of wave tiles at a time)
constexpr int wgrad_lds_floats(int NT)
{
    const int QT = wgrad_qt(NT);
    const int WMk = wgrad_wmk(QT), WNk = wgrad_wnk(QT), WK = wgrad_wk(QT);
    const int NWv = WMk * WNk, TN = (QT + WNk - 1) / WNk;
    const int fold = (WK - 1) * NWv * TN * 1024 + (WK - 1) * NWv * 64, panels = 4 * 32 * QT * 36;
    return fold > panels ? fold : panels;
}

// NT > 0: compile-time width.  NT == 0: run-time width above 16 tiles (k_lean's nets), QTR tiles per quadrant side, ceil(nt / 8)
// quadrants per side; the last quadrant row / column may hold fewer than QTR tiles (its missing rows are staged as zeros).
template <int NT, int QTR = 0>
__global__ __launch_bounds__(512, 2) void k_wgrad(const WgradArgs a)
{
    constexpr bool RTW = NT == 0;
    const int nt_w = RTW ? brief_nt(a.d) : NT;
    const int FP = 32 * nt_w;
    const int NQ = RTW ? (nt_w + 7) / 8 : wgrad_nq(NT);
    constexpr int QT = RTW ? QTR : wgrad_qt(NT), QP = 32 * QT;   // quadrants per side, tiles / rows per quadrant side
    // 8 waves = WMk x WNk output-tile grid x WK-way split of each chunk's four k-groups.  Wide nets spend
    // all 8 waves on output tiles; narrow ones (QT <= 4) would leave most waves without a tile, so they
    // split K instead and fold the partial accumulators through LDS at the end (fixed order).
    // LIST (5 tiles per side: 25 tiles): wave w owns tiles [3 w, 3 w + 3) of the block in row-major order, wave 7 four of them — 6, 6, 6 and 7
    // tiles on the four SIMDs where 2 x 2 waves x (3 x 3) x two k-slices carried 9; both fragments of a tile are read per tile (LDS has the room).
    constexpr bool LIST = wgrad_list(QT);
    constexpr int LCNT = QT * QT / 8, LREM = QT * QT % 8;      // tiles per wave; the last LREM waves (on different SIMDs, LREM <= 4) take one more
    constexpr bool HET7 = QT == 7 && !LIST;
    // 7 tiles per side (HET7): no rectangular grid of equal wave tiles covers 49 tiles with fewer than 64 slots, so the waves get UNEQUAL
    // pieces — wave w < 7 the first six tiles of tile row w, wave 7 the whole tile column 6: 12, 12, 12 and 13 tiles on the four SIMDs (a SIMD
    // carries waves w and w + 4) instead of 16, and at most 7 accumulator tiles per wave.
    constexpr int WMk = HET7 || LIST ? 2 : wgrad_wmk(QT), WNk = HET7 || LIST ? 4 : wgrad_wnk(QT);      // (HET7, LIST: all 8 waves in k-slice 0)
    constexpr int WK = wgrad_wk(QT);
    constexpr int NWv = WMk * WNk;                 // waves per k-slice
    constexpr int TM = HET7 || LIST ? 1 : (QT + WMk - 1) / WMk, TN = LIST ? LCNT + (LREM ? 1 : 0) : (HET7 ? 7 : (QT + WNk - 1) / WNk);
    constexpr bool MEX = QT % WMk == 0, NEX = QT % WNk == 0;   // every wave tile exists
    constexpr int LDSW = 36;                       // row stride (floats): conflict-free ds_read_b128
    constexpr int NLD = (QP * 8 + 511) / 512;      // float4 loads per thread per operand per chunk
    constexpr int PANEL = QP * LDSW;
    constexpr bool FULL = (QP * 8) % 512 == 0;     // every thread has a slot in every staging pass
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2 buffers][A panel | B panel]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, ln = lane & 31;
    const int wk = wave / NWv, wrem = wave % NWv;
    const int wmk = wrem / WNk, wnk = wrem % WNk;
    const bool kactive = wk < WK;                  // QT == 1 uses 4 of the 8 waves for MFMA work
    const int quad = blockIdx.x % (NQ * NQ), bsl = blockIdx.x / (NQ * NQ);
    const int qm = quad / NQ, qn = quad % NQ;
    const int l = 1 + bsl / a.nsplit;              // hidden layer 1..L-2
    const int split = bsl % a.nsplit;
    const int64_t nchunks = a.npad / 32;
    const int64_t c0 = nchunks * split / a.nsplit, c1 = nchunks * (split + 1) / a.nsplit;   // c1 > c0 (host: nsplit <= nchunks)
    // stash planes are [32-sample chunk][FP rows][32 samples]: this quadrant's rows of a chunk are QP * 128 contiguous bytes
    const float *Dl = a.D + (int64_t)(l - 1) * FP * a.npad + qm * QP * 32;     // this quadrant's delta rows (of chunk 0)
    const float *Zl = a.Z + (int64_t)(l - 1) * FP * a.npad + qn * QP * 32;     // ... and phase rows

    f32x16 acc[TM][TN];
    float dbacc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        dbacc[i] = 0.f;
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;
    }
    float dbl[LIST ? TN : 1];       // LIST: row sums of every tile's A fragments (the bias gradient where the tile sits in column 0)
    int lmt[LIST ? TN : 1], lnt[LIST ? TN : 1];
    if (LIST) {
        const int first = wave * LCNT + (wave > 8 - LREM ? wave - (8 - LREM) : 0);
#pragma unroll
        for (int k = 0; k < TN; ++k) {
            const int t = first + k < QT * QT ? first + k : QT * QT - 1;
            lmt[k] = t / QT; lnt[k] = t % QT; dbl[k] = 0.f;
        }
    }
    float4 ra[NLD], rb[NLD];
    // The loop body is branch-free: the prefetch of "chunk c+2" and the staging of "chunk c+1" are
    // clamped to the last chunk instead of being skipped (the redundant copies are never read).
    // thread e fetches 16 bytes at e * 16 of the quadrant's rows; the chunk offset is a scalar (no per-load 64-bit address
    // arithmetic on the VALU, which the f32 MFMA shares)
    const int panel_bytes = (int)((int64_t)FP * a.npad * 4) - (RTW ? 0 : (NQ - 1) * QP * 128);      // to the end of the last chunk's rows of the last quadrant
    const int rows_m = RTW ? FP - qm * QP : QP, rows_n = RTW ? FP - qn * QP : QP;      // rows this quadrant really has (run-time width: the last one may be short)
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void *)Dl, 0, panel_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc((void *)Zl, 0, panel_bytes, 0x00020000);
    int voffs[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + 512 * i;
        voffs[i] = e * 16;
    }
#define WG_ISSUE(cc)                                                                              \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                             \
        const int e = tid + 512 * i;                                                              \
        ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);                                                  \
        rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);                                                  \
        if (FULL || e < QP * 8) {                                                                 \
            if (!RTW || (e >> 3) < rows_m) ra[i] = bload4w(rsD, voffs[i], (int)((cc) * (FP * 128)));     \
            if (!RTW || (e >> 3) < rows_n) rb[i] = bload4w(rsZ, voffs[i], (int)((cc) * (FP * 128)));     \
        }                                                                                         \
    }
#define WG_STAGE_A(buf, i)                                                                        \
    {                                                                                             \
        const int e = tid + 512 * (i);                                                            \
        if (FULL || e < QP * 8)                                                                   \
            *reinterpret_cast<float4 *>(smem + (buf) * 2 * PANEL + (e >> 3) * LDSW + (e & 7) * 4) = ra[i]; \
    }
#define WG_STAGE_B(buf, i)                                                                        \
    {                                                                                             \
        const int e = tid + 512 * (i);                                                            \
        if (FULL || e < QP * 8) {                                                                 \
            float4 h;                                                                             \
            h.x = BRIEF_SIN_REV(rb[i].x); h.y = BRIEF_SIN_REV(rb[i].y);   /* the stash holds revolutions */ \
            h.z = BRIEF_SIN_REV(rb[i].z); h.w = BRIEF_SIN_REV(rb[i].w);                           \
            *reinterpret_cast<float4 *>(smem + (buf) * 2 * PANEL + PANEL + (e >> 3) * LDSW + (e & 7) * 4) = h; \
        }                                                                                         \
    }
    WG_ISSUE(c0)
#pragma unroll
    for (int i = 0; i < NLD; ++i) { WG_STAGE_A(0, i) WG_STAGE_B(0, i) }
    {
        const int64_t cn = c0 + 1 < c1 ? c0 + 1 : c1 - 1;
        WG_ISSUE(cn)
    }
    lds_barrier();
#ifdef BRIEF_STAMPS
    float st_acc[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    long long st_last = clock64();
#endif
    // (HET7: one copy of the whole chunk loop per role — a role test INSIDE the loop makes the compiler carry two sets of accumulators
    //  through the merge points and spill; every wave meets the same barriers in either copy)
    auto chunk_loop = [&](auto role_c) __attribute__((always_inline)) {
    constexpr int ROLE = decltype(role_c)::value;
    for (int64_t c = c0; c < c1; ++c) {
        const int cur = (int)(c - c0) & 1;
        const float *As = smem + cur * 2 * PANEL, *Bs = As + PANEL;
        STAMP(0)
        if constexpr (LIST) {
            constexpr int CNT = LCNT + ROLE;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
#pragma unroll
                for (int k = 0; k < CNT; ++k) {
                    const float4 a_ = *reinterpret_cast<const float4 *>(As + (32 * lmt[k] + ln) * LDSW + 8 * gq + 4 * hi);
                    const float4 b_ = *reinterpret_cast<const float4 *>(Bs + (32 * lnt[k] + ln) * LDSW + 8 * gq + 4 * hi);
                    dbl[k] += (a_.x + a_.y) + (a_.z + a_.w);
                    acc[0][k] = MFMA(a_.x, b_.x, acc[0][k]); acc[0][k] = MFMA(a_.y, b_.y, acc[0][k]);
                    acc[0][k] = MFMA(a_.z, b_.z, acc[0][k]); acc[0][k] = MFMA(a_.w, b_.w, acc[0][k]);
                }
                if (gq >= 2) {
#pragma unroll
                    for (int i = 0; i < NLD; ++i)
                        if ((i & 1) == (gq & 1)) { WG_STAGE_A(cur ^ 1, i) WG_STAGE_B(cur ^ 1, i) }
                }
            }
        } else if constexpr (HET7) {
            // two wave-uniform roles, each a straight-line stream of its own over the chunk's four k-groups (the role test sits OUTSIDE
            // the k-group loop so that each stream keeps its own software pipeline of fragment reads under the MFMAs)
#define WG_FRAG(P, row) (*reinterpret_cast<const float4 *>((P) + (32 * (row) + ln) * LDSW + 8 * gq + 4 * hi))
#define WG_MFMA4(k, A_, B_) { acc[0][k] = MFMA((A_).x, (B_).x, acc[0][k]); acc[0][k] = MFMA((A_).y, (B_).y, acc[0][k]); \
                              acc[0][k] = MFMA((A_).z, (B_).z, acc[0][k]); acc[0][k] = MFMA((A_).w, (B_).w, acc[0][k]); }
#define WG_STAGE_NEXT                                                                   \
            if (gq >= 2) {                                                              \
                _Pragma("unroll") for (int i = 0; i < NLD; ++i)                         \
                    if ((i & 1) == (gq & 1)) { WG_STAGE_A(cur ^ 1, i) WG_STAGE_B(cur ^ 1, i) } \
            }
            if constexpr (ROLE == 0) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 a_ = WG_FRAG(As, wave);
                    dbacc[0] += (a_.x + a_.y) + (a_.z + a_.w);
#pragma unroll
                    for (int jn = 0; jn < 6; ++jn) {
                        const float4 b_ = WG_FRAG(Bs, jn);
                        WG_MFMA4(jn, a_, b_)
                    }
                    WG_STAGE_NEXT
                }
            } else {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 b6 = WG_FRAG(Bs, 6);
#pragma unroll
                    for (int i = 0; i < 7; ++i) {
                        const float4 a_ = WG_FRAG(As, i);
                        WG_MFMA4(i, a_, b6)
                    }
                    WG_STAGE_NEXT
                }
            }
#undef WG_FRAG
#undef WG_MFMA4
#undef WG_STAGE_NEXT
        } else
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            if (kactive && (WK == 1 || (gq % WK) == wk)) {
                float4 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int mt = wmk * TM + i;
                    af[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (MEX || mt < QT) af[i] = *reinterpret_cast<const float4 *>(As + (32 * mt + ln) * LDSW + 8 * gq + 4 * hi);
                    if (wnk == 0) dbacc[i] += (af[i].x + af[i].y) + (af[i].z + af[i].w);
                }
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
                    const int nt = wnk * TN + jn;
                    bf[jn] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (NEX || nt < QT) bf[jn] = *reinterpret_cast<const float4 *>(Bs + (32 * nt + ln) * LDSW + 8 * gq + 4 * hi);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn) {
                        if ((MEX || wmk * TM + i < QT) && (NEX || wnk * TN + jn < QT)) {
                            acc[i][jn] = MFMA(af[i].x, bf[jn].x, acc[i][jn]);
                            acc[i][jn] = MFMA(af[i].y, bf[jn].y, acc[i][jn]);
                            acc[i][jn] = MFMA(af[i].z, bf[jn].z, acc[i][jn]);
                            acc[i][jn] = MFMA(af[i].w, bf[jn].w, acc[i][jn]);
                        }
                    }
                }
            }
            // The next chunk's staging rides in the shadow of the LAST two MFMA groups: its global loads were
            // issued at the end of the previous iteration and get the barrier + two groups to land.
            if (gq >= 2) {
#pragma unroll
                for (int i = 0; i < NLD; ++i)
                    if ((i & 1) == (gq & 1)) { WG_STAGE_A(cur ^ 1, i) WG_STAGE_B(cur ^ 1, i) }
            }
        }
        STAMP(1)
        {
            const int64_t cn = c + 2 < c1 ? c + 2 : c1 - 1, cn1 = c + 1 < c1 ? c + 1 : c1 - 1;
            WG_ISSUE(cn)
        }
        STAMP(2)
        lds_barrier();
        STAMP(3)
    }
    };
    if (LIST ? (LREM > 0 && wave >= 8 - LREM) : (HET7 && wave == 7)) chunk_loop(RoleC<1>{});
    else chunk_loop(RoleC<0>{});
#ifdef BRIEF_STAMPS
    if (lane == 0 && a.stamps) for (int i = 0; i < 4; ++i) a.stamps[((int64_t)blockIdx.x * 8 + wave) * 8 + i] = st_acc[i];
#endif
#undef WG_ISSUE
#undef WG_STAGE_A
#undef WG_STAGE_B
    if (WK > 1) {
        // fold the k-slices, one row of wave tiles at a time: slices 1..WK-1 park accumulator row i in LDS (the panels are dead now:
        // the loop ended on a barrier), slice 0 adds them in slice order (row by row so that the area never exceeds the panels)
        float *red = smem;
        float *redb = smem + (WK - 1) * NWv * TN * 1024;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (i > 0) lds_barrier();      // slice 0 is done with the previous row's partials
            if (kactive && wk > 0) {
#pragma unroll
                for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        red[(((wk - 1) * NWv + wrem) * TN + jn) * 1024 + r * 64 + lane] = acc[i][jn][r];
                redb[((wk - 1) * NWv + wrem) * 64 + lane] = dbacc[i];
            }
            lds_barrier();
            if (wk == 0) {
                for (int w = 1; w < WK; ++w) {
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            acc[i][jn][r] += red[(((w - 1) * NWv + wrem) * TN + jn) * 1024 + r * 64 + lane];
                    dbacc[i] += redb[((w - 1) * NWv + wrem) * 64 + lane];
                }
            }
        }
    }
    if (wk != 0) return;
    float *slab = a.slabs + ((int64_t)(l - 1) * a.nsplit + split) * ((int64_t)FP * FP + FP);
    if (LIST) {
        const int cnt = LCNT + (LREM > 0 && wave >= 8 - LREM ? 1 : 0);
#pragma unroll
        for (int k = 0; k < TN; ++k) {
            if (k >= cnt) break;
            const int mt = lmt[k], nt = lnt[k];
            if ((!RTW || qm * QT + mt < nt_w) && (!RTW || qn * QT + nt < nt_w)) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    slab[(int64_t)(qm * QP + 32 * mt + ROWMAP(r, hi)) * FP + qn * QP + 32 * nt + ln] = acc[0][k][r];
            }
            if (qn == 0 && nt == 0) {
                const float tot = dbl[k] + __shfl_xor(dbl[k], 32);
                if (hi == 0 && (!RTW || qm * QT + mt < nt_w)) slab[(int64_t)FP * FP + qm * QP + 32 * mt + ln] = tot;
            }
        }
        return;
    }
    if (HET7) {
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int mt = wave < 7 ? wave : k, nt = wave < 7 ? k : 6;
            if (wave < 7 && k == 6) continue;
            if ((!RTW || qm * QT + mt < nt_w) && (!RTW || qn * QT + nt < nt_w)) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    slab[(int64_t)(qm * QP + 32 * mt + ROWMAP(r, hi)) * FP + qn * QP + 32 * nt + ln] = acc[0][k][r];
            }
        }
        if (qn == 0 && wave < 7) {      // bias gradients: tile row w from wave w
            const float tot = dbacc[0] + __shfl_xor(dbacc[0], 32);
            if (hi == 0 && (!RTW || qm * QT + wave < nt_w)) slab[(int64_t)FP * FP + qm * QP + 32 * wave + ln] = tot;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mt = wmk * TM + i;
        if (mt < QT && (!RTW || qm * QT + mt < nt_w)) {
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                const int nt = wnk * TN + jn;
                if (nt < QT && (!RTW || qn * QT + nt < nt_w)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        slab[(int64_t)(qm * QP + 32 * mt + ROWMAP(r, hi)) * FP + qn * QP + 32 * nt + ln] = acc[i][jn][r];
                }
            }
            if (wnk == 0 && qn == 0) {
                const float tot = dbacc[i] + __shfl_xor(dbacc[i], 32);
                if (hi == 0) slab[(int64_t)FP * FP + qm * QP + 32 * mt + ln] = tot;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// BRIEF_PREC_BF16X3 weight-gradient GEMM (FP = 256): the same split-K job as k_wgrad<8> on the same f32 stashes, with both
// operands split into hi + lo bf16 halves while they are staged (h = sin(2 pi phase) first) and three v_mfma_f32_32x32x16_bf16
// per fragment pair.  Panel rows are [32 samples hi | 32 samples lo | pad] = 144 B, the f32 kernel's conflict-free 36-dword
// stride; a fragment is 16 B of a row (8 consecutive samples).  Bias gradients: row sums of the f32 deltas, taken by the threads
// that stage them.  Same slab format as k_wgrad: k_reduce does not know the difference.
#ifndef BRIEF_X3W_DLY
#define BRIEF_X3W_DLY 0
#endif
__global__ __launch_bounds__(512, 2) void k_wgrad_x3(const WgradArgs a)
{
    constexpr int FP = 256, QT = 8, WNk = 4, TM = 4, TN = 2, LDSW = 36, NLD = 4, PANEL = FP * LDSW;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2 buffers][A panel | B panel]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, ln = lane & 31;
    const int wmk = wave / WNk, wnk = wave % WNk;
    const int l = 1 + (int)blockIdx.x / a.nsplit;                  // hidden layer 1..L-2
    const int split = (int)blockIdx.x % a.nsplit;
    const int64_t nchunks = a.npad / 32;
    const int64_t c0 = nchunks * split / a.nsplit, c1 = nchunks * (split + 1) / a.nsplit;
    const float *Dl = a.D + (int64_t)(l - 1) * FP * a.npad;
    const float *Zl = a.Z + (int64_t)(l - 1) * FP * a.npad;
    (void)QT;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;
    float bsum[NLD] = {0.f, 0.f, 0.f, 0.f};
    u32x4 ra[NLD], rb[NLD];      // kept as whole 128-bit tuples: as float4 structs the loop-carried components were shuffled
                                 // through copies behind s_waitcnt at the loop end, i.e. no prefetch across iterations
#ifdef BRIEF_X3W_NOLOAD      // diagnostic builds (timing only, results are garbage): no memory traffic / no MFMAs / no staging
    const int panel_bytes = 0;
#else
    const int panel_bytes = (int)((int64_t)FP * a.npad * 4);
#endif
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void *)Dl, 0, panel_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc((void *)Zl, 0, panel_bytes, 0x00020000);
    int voffs[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + 512 * i;
        voffs[i] = e * 16;
    }
#define X3F(u_) __uint_as_float(u_)
#ifndef BRIEF_X3W_AUX
#define BRIEF_X3W_AUX 2      // streaming cache policy for the operand panels: read once (aux 0: 155 us, aux 2: 115 us)
#endif
#define X3W_LD(rs_, base_, voff_, soff_) __builtin_amdgcn_raw_buffer_load_b128(rs_, voff_, soff_, BRIEF_X3W_AUX)
#define X3W_ISSUE(cc)                                                                             \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                             \
        ra[i] = X3W_LD(rsD, Dl, voffs[i], (int)((cc) * (FP * 128)));       \
        rb[i] = X3W_LD(rsZ, Zl, voffs[i], (int)((cc) * (FP * 128)));       \
    }
    // four f32 values of one row -> 4 hi + 4 lo bf16 (8 + 8 bytes) in the row's hi / lo halves
#define X3W_PUT(dst_, v_)                                                                         \
    {                                                                                             \
        union { uint2 u; __bf16 h[4]; } ph_, pl_;                                                 \
        const float f_[4] = {__uint_as_float((v_).x), __uint_as_float((v_).y), __uint_as_float((v_).z), __uint_as_float((v_).w)}; \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                        \
            ph_.h[j_] = (__bf16)f_[j_];                                                           \
            pl_.h[j_] = (__bf16)(f_[j_] - (float)ph_.h[j_]);                                      \
        }                                                                                         \
        *reinterpret_cast<uint2 *>(dst_) = ph_.u;                                                 \
        *reinterpret_cast<uint2 *>((dst_) + 16) = pl_.u;       /* + 64 bytes: the lo half */      \
    }
#define X3W_STAGE(buf, count_)                                                                    \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                             \
        const int e = tid + 512 * i;                                                              \
        float *pa_ = smem + (buf) * 2 * PANEL + (e >> 3) * LDSW + (e & 7) * 2;                    \
        X3W_PUT(pa_, ra[i])                                                                       \
        if (count_) bsum[i] += (X3F(ra[i].x) + X3F(ra[i].y)) + (X3F(ra[i].z) + X3F(ra[i].w));     \
        u32x4 h_;                                                                                 \
        h_.x = __float_as_uint(BRIEF_SIN_REV(X3F(rb[i].x))); h_.y = __float_as_uint(BRIEF_SIN_REV(X3F(rb[i].y))); \
        h_.z = __float_as_uint(BRIEF_SIN_REV(X3F(rb[i].z))); h_.w = __float_as_uint(BRIEF_SIN_REV(X3F(rb[i].w))); \
        X3W_PUT(pa_ + PANEL, h_)                                                                  \
    }
    X3W_ISSUE(c0)
    X3W_STAGE(0, true)
    {
        const int64_t cn = c0 + 1 < c1 ? c0 + 1 : c1 - 1;
        X3W_ISSUE(cn)
    }
    lds_barrier();
    for (int64_t c = c0; c < c1; ++c) {
        const int cur = (int)(c - c0) & 1;
        const float *As = smem + cur * 2 * PANEL, *Bs = As + PANEL;
        // Chunk c + 1 is staged (its loads were issued an iteration ago) and chunk c + 2 requested BETWEEN this chunk's MFMA groups:
        // a bf16 MFMA hides ~4 VALU instructions of its own wave (profiles/r03_coissue.md), so the 16 groups of three MFMAs
        // carry the 4 x 4 staging sub-steps (pack delta | sines | pack h | next loads) instead of a separate VALU phase.  The
        // staged buffer is the one nobody reads this iteration.
        const float flag = c + 1 < c1 ? 1.0f : 0.0f;        // the clamped re-staging of the last chunk does not count in the bias sums
        const int64_t cn = c + 2 < c1 ? c + 2 : c1 - 1, cn1 = c + 1 < c1 ? c + 1 : c1 - 1;
        u32x4 hs = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // A fragments one row-tile ahead of their MFMAs (2 x 8 registers), B fragments for the whole k-step (16): with all four row
            // tiles' fragments resident (32) the allocator parked two of the prefetched operand tuples elsewhere and copied them back
            // behind s_waitcnt vmcnt(3) at the end of every iteration
            X3Frag ah[2], al[2], bh[TN], bl[TN];
            {
                const float *p_ = As + (32 * (wmk * TM) + ln) * LDSW + 8 * ks + 4 * hi;           // dwords: (16 ks + 8 hi) samples x 2 B
                ah[0].u = *reinterpret_cast<const uint4 *>(p_);
                al[0].u = *reinterpret_cast<const uint4 *>(p_ + 16);
            }
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                const float *p_ = Bs + (32 * (wnk * TN + jn) + ln) * LDSW + 8 * ks + 4 * hi;
                bh[jn].u = *reinterpret_cast<const uint4 *>(p_);
                bl[jn].u = *reinterpret_cast<const uint4 *>(p_ + 16);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
                    if (jn == 0 && i + 1 < TM) {
                        const float *p_ = As + (32 * (wmk * TM + i + 1) + ln) * LDSW + 8 * ks + 4 * hi;
                        ah[(i + 1) & 1].u = *reinterpret_cast<const uint4 *>(p_);
                        al[(i + 1) & 1].u = *reinterpret_cast<const uint4 *>(p_ + 16);
                    }
#ifndef BRIEF_X3W_NOMFMA
                    acc[i][jn] = MFMA_X3(ah[i & 1].v, bh[jn].v, acc[i][jn]);
                    acc[i][jn] = MFMA_X3(ah[i & 1].v, bl[jn].v, acc[i][jn]);
                    acc[i][jn] = MFMA_X3(al[i & 1].v, bh[jn].v, acc[i][jn]);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    const int g = ks * (TM * TN) + i * TN + jn, slot = g >> 2, sub = g & 3;      // compile-time after unrolling
                    const int e = tid + 512 * slot;
                    float *pa_ = smem + (cur ^ 1) * 2 * PANEL + (e >> 3) * LDSW + (e & 7) * 2;
#ifndef BRIEF_X3W_NOSTAGE
                    // (the empty asm statements make the loop-carried values the load tuples themselves: left to the SLP vectoriser they
                    //  were component pairs across slots, copied into place behind s_waitcnt vmcnt(0) at the end of every iteration)
                    if (sub == 0) {
                        asm volatile("" : "+v"(ra[slot]));
                        X3W_PUT(pa_, ra[slot])
                        // (kept scalar and opaque: vectorised across slots, the sums were formed in place in the registers of
                        //  ra[0] / ra[2], whose reloads then went elsewhere and were copied back behind s_waitcnt vmcnt(3) at
                        //  the end of every iteration — five of the eight loads just issued had to land before the next chunk)
                        float rs_ = (X3F(ra[slot].x) + X3F(ra[slot].y)) + (X3F(ra[slot].z) + X3F(ra[slot].w));
                        asm volatile("" : "+v"(rs_));
                        bsum[slot] += flag * rs_;
                    } else if (sub == 1) {
                        asm volatile("" : "+v"(rb[slot]));
                        hs.x = __float_as_uint(BRIEF_SIN_REV(X3F(rb[slot].x))); hs.y = __float_as_uint(BRIEF_SIN_REV(X3F(rb[slot].y)));
                        hs.z = __float_as_uint(BRIEF_SIN_REV(X3F(rb[slot].z))); hs.w = __float_as_uint(BRIEF_SIN_REV(X3F(rb[slot].w)));
                    } else if (sub == 2) {
                        X3W_PUT(pa_ + PANEL, hs)
                    }
                    // the freed slot is re-requested BRIEF_X3W_DLY groups later (wrapping into the next iteration, ahead of the slot's
                    // staging there): the memory system delivers more with less in flight (tools/hbm_read_ubench.hip: 6.4 TB/s at
                    // 16-32 KB per CU, 5.5 at 64 KB and above)
                    {
                        const int gl = (g + 16 - 3 - BRIEF_X3W_DLY) & 15;      // the group whose slot is due now (compile-time after unrolling)
                        if ((gl & 3) == 0) {
                            const int sl = gl >> 2;
                            const int64_t cl = (4 * sl + 3 + BRIEF_X3W_DLY < 16) ? cn : cn1;
                            ra[sl] = X3W_LD(rsD, Dl, voffs[sl], (int)(cl * (FP * 128)));
                            rb[sl] = X3W_LD(rsZ, Zl, voffs[sl], (int)(cl * (FP * 128)));
                        }
                    }
#else
                    if (sub == 3) {
                        float rs_ = X3F(ra[slot].x) + X3F(rb[slot].x);
                        asm volatile("" : "+v"(rs_));
                        bsum[slot] += flag * rs_;
                        ra[slot] = X3W_LD(rsD, Dl, voffs[slot], (int)(cn * (FP * 128)));
                        rb[slot] = X3W_LD(rsZ, Zl, voffs[slot], (int)(cn * (FP * 128)));
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        lds_barrier();
    }
#undef X3W_ISSUE
#undef X3W_LD
#undef X3F
#undef X3W_PUT
#undef X3W_STAGE
    float *slab = a.slabs + ((int64_t)(l - 1) * a.nsplit + split) * ((int64_t)FP * FP + FP);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mt = wmk * TM + i;
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int nt = wnk * TN + jn;
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(int64_t)(32 * mt + ROWMAP(r, hi)) * FP + 32 * nt + ln] = acc[i][jn][r];
        }
    }
    // bias gradients: the eight threads that staged the eight 16-byte pieces of a row are neighbouring lanes
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        float v = bsum[i];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
        if ((tid & 7) == 0) slab[(int64_t)FP * FP + (tid >> 3) + 64 * i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// optimizer: same arithmetic, in the same order, as oracle_optim_step (torch single-tensor rules)
struct OptimScalars { int kind; float w1, fb2, w2, feps, nstep, bc2s; };

__device__ __forceinline__ float optim_apply(const OptimScalars &o, float gi, float *__restrict__ p, float *__restrict__ s1,
                                             float *__restrict__ s2, int64_t i)
{
    const int kind = o.kind;
    const float w1 = o.w1, fb2 = o.fb2, w2 = o.w2, feps = o.feps, nstep = o.nstep, bc2s = o.bc2s;
    if (kind == BRIEF_OPT_ADAMAX) {
        const float m = __fmaf_rn(w1, __fsub_rn(gi, s1[i]), s1[i]);
        const float ua = __fmul_rn(s2[i], fb2), ub = __fadd_rn(fabsf(gi), feps);
        const float u = ua > ub ? ua : ub;
        s1[i] = m; s2[i] = u;
        p[i] = __fadd_rn(p[i], __fdiv_rn(__fmul_rn(nstep, m), u));
    } else if (kind == BRIEF_OPT_ADAM) {
        const float m = __fmaf_rn(w1, __fsub_rn(gi, s1[i]), s1[i]);
        const float v = __fadd_rn(__fmul_rn(s2[i], fb2), __fmul_rn(w2, __fmul_rn(gi, gi)));
        s1[i] = m; s2[i] = v;
        const float den = __fadd_rn(__fdiv_rn(sqrtf(v), bc2s), feps);   // sqrtf: correctly rounded (hipcc default)
        p[i] = __fadd_rn(p[i], __fdiv_rn(__fmul_rn(nstep, m), den));
    } else {
        p[i] = __fadd_rn(p[i], __fmul_rn(nstep, gi));
    }
    return p[i];
}

__global__ void k_optim(OptimScalars o, float *__restrict__ p, const float *__restrict__ g, float *__restrict__ s1,
                        float *__restrict__ s2, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    optim_apply(o, g[i], p, s1, s2, i);
}

// ---------------------------------------------------------------------------------------------
// deterministic reduction of the partials into the canonical gradient buffer
struct ReduceArgs {
    brief_siren_desc d;
    const float *rec;     // [nrec_wg*4][REC]
    int nrec_wg;
    const float *slabs;
    int nsplit;
    int sgroups;          // hidden parameters: threads per parameter (4 for k_wgrad's <= 85 slabs, 8 for k_small's one per workgroup)
    float *grads;
    float *loss_out;
    float inv_count;
    // optional fused optimizer step + scatter into the fragment-ordered copies (brief_siren_fit_step)
    int update;
    OptimScalars opt;
    float *params, *s1, *s2, *pk;
};

// where a hidden-layer weight W_l[o][i] lives in the A-fragment orders (see brief_layout.h)
__device__ __forceinline__ int64_t frag_index(int NT, int row, int col)
{
    const int mt = row >> 5, ii = row & 31, kt = col >> 5, rem = col & 31;
    const int q = rem >> 3, hi = (rem >> 2) & 1, j = rem & 3;
    return ((((int64_t)mt * NT + kt) * 4 + q) * 64 + 32 * hi + ii) * 4 + j;
}

// blocks [0, nb_hidden): one thread per hidden-layer parameter, nsplit slab terms each.
// blocks [nb_hidden, ...): one WAVE per first-layer / head parameter (and one for the loss): these sum
// over up to 2048 per-wave records, which a single thread would walk at one L2 latency per term.
// bid: this block's number among the blocks that serve THIS job (the whole grid for k_reduce); opt / loss_out: the step's optimizer
// scalars and loss destination (per-step values: kernel arguments also in the grouped form)
__device__ __forceinline__ void reduce_body(const ReduceArgs &a, const OptimScalars &opt, float *loss_out, int nb_hidden, int bid)
{
    const brief_siren_desc &d = a.d;
    const int F = d.features, cin = d.cin, cout = d.cout;
    const int NT = brief_nt(d), FP = 32 * NT, WM = brief_wm(NT), WS = brief_ws(NT);
    const int64_t off_head = brief_canon_head_off(d);
    const int64_t l0_count = (int64_t)F * cin + F;
    if (bid < nb_hidden) {
        // sgroups threads per parameter: thread (pl, sg) adds slabs sg, sg + sgroups, ...; group 0 then adds the
        // group sums in group order (fixed order: bit-reproducible)
        __shared__ float fold[1024];
        const int64_t hcount = off_head - l0_count;
        const int SG = a.sgroups, ppb = (int)blockDim.x / SG;      // ppb consecutive parameters per block: a group's loads stay coalesced
        const int pl = threadIdx.x % ppb, sg = threadIdx.x / ppb;
        const int64_t hidx = (int64_t)bid * ppb + pl;
        const bool live = hidx < hcount;
        const int64_t per = (int64_t)F * F + F;
        const int l = live ? 1 + (int)(hidx / per) : 1;
        const int64_t r = live ? hidx % per : 0;
        float s = 0.f;
        if (live) {
            int64_t so;
            if (r < (int64_t)F * F) so = (r / F) * FP + (r % F);
            else so = (int64_t)FP * FP + (r - (int64_t)F * F);
            const int64_t slab_sz = (int64_t)FP * FP + FP;
            const float *base = a.slabs + (int64_t)(l - 1) * a.nsplit * slab_sz + so;
            if (SG >= 16) {
                // k_small's one slab per workgroup: <= 16 terms per thread, all loads in flight at once
#pragma unroll 16
                for (int sp = sg; sp < a.nsplit; sp += SG) s += base[(int64_t)sp * slab_sz];
            } else {
#pragma unroll 8
                for (int sp = sg; sp < a.nsplit; sp += SG) s += base[(int64_t)sp * slab_sz];
            }
        }
        if (SG > 1) {
            fold[threadIdx.x] = s;
            __syncthreads();
            if (sg == 0)
                for (int gq = 1; gq < SG; ++gq) s += fold[gq * ppb + pl];
        }
        if (!live || sg != 0) return;
        a.grads[l0_count + hidx] = s;
        if (a.update) {
            const float pv = optim_apply(opt, s, a.params, a.s1, a.s2, l0_count + hidx);
            float *blk = a.pk + brief_pk_hidden(d, l);
            if (r < (int64_t)F * F) {
                const int o = (int)(r / F), i = (int)(r % F);
                // the same products as k_repack's, bit for bit (brief_layout.h: what the copies carry)
                const float pf = brief_phase_scale(d, l) * pv, pb = brief_om_prev(d, l) * pv;
                if (d.precision != BRIEF_PREC_BF16) {
                    blk[frag_index(NT, o, i)] = pf;                               // A-fragments of s_l W
                    blk[(int64_t)FP * FP + frag_index(NT, i, o)] = pb;            // A-fragments of w0_{l-1} W^T
                }
                if (d.precision == BRIEF_PREC_BF16X3) {
                    // hi + lo halves of the same two products (k_repack forms them the same way)
                    uint16_t *h16 = reinterpret_cast<uint16_t *>(a.pk + brief_pk16_off(d, l));
                    uint16_t *l16 = reinterpret_cast<uint16_t *>(a.pk + brief_pk16_off(d, l) + brief_pk16_region(d));
                    uint16_t fh, fl, bh, bl;
                    x3_split_f16(BRIEF_X3_FWD_SCALE * pf, fh, fl);      // forward copy: fp16 halves of 2^6 om W / 2 pi
                    x3_split_bf16(pb, bh, bl);                          // backward copy: bf16 halves of om W^T
                    h16[brief_frag16_index(NT, o, i)] = fh;
                    h16[(int64_t)FP * FP + brief_frag16_index(NT, i, o)] = bh;
                    l16[brief_frag16_index(NT, o, i)] = fl;
                    l16[(int64_t)FP * FP + brief_frag16_index(NT, i, o)] = bl;
                } else if (d.precision == BRIEF_PREC_BF16) {
                    // the bf16 kernels read only the bf16 fragments below (and the f32 first layer, biases and head): the f32
                    // hidden fragment slots hold zeros (k_repack) and are never read in this mode — two scattered 4-byte stores
                    // per parameter less
                    __bf16 *b16 = reinterpret_cast<__bf16 *>(a.pk + brief_pk16_off(d, l));
                    b16[brief_frag16_index(NT, o, i)] = (__bf16)pf;
                    b16[(int64_t)FP * FP + brief_frag16_index(NT, i, o)] = (__bf16)pb;
                }
            } else {
                blk[2 * (int64_t)FP * FP + (r - (int64_t)F * F)] = brief_phase_scale(d, l) * pv;        // bias
            }
        }
        return;
    }
    // ---- skinny parameters: wave w of this block handles item (blockIdx - nb_hidden)*4 + w
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)(bid - nb_hidden) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t head_count = (int64_t)cout * F + cout;
    const int nrec = a.nrec_wg * WS;                 // records that carry a given feature tile / the loss
    const int TR = brief_rec_tr(NT), rec_floats = brief_rec_floats(NT);      // record sections: dW0[TR][4] | dWh[4][TR] | dbh[4] | loss
    if (item > l0_count + head_count) return;
    int slot, wmo = 0;
    if (item == l0_count + head_count) {
        slot = 8 * TR + 4;                            // loss (waves with wm == 0)
    } else if (item < l0_count) {
        int o, c;
        if (item < (int64_t)F * cin) { o = (int)(item / cin); c = (int)(item % cin); }
        else { o = (int)(item - (int64_t)F * cin); c = 3; }
        slot = (((o >> 5) / WM) * 32 + (o & 31)) * 4 + c;
        wmo = (o >> 5) % WM;
    } else {
        const int64_t r = item - l0_count;
        if (r < (int64_t)cout * F) {
            const int c = (int)(r / F), o = (int)(r % F);
            slot = 4 * TR + c * TR + ((o >> 5) / WM) * 32 + (o & 31);
            wmo = (o >> 5) % WM;
        } else {
            slot = 8 * TR + (int)(r - (int64_t)cout * F);
        }
    }
    // record index of (wg, ws) for this wm: (wg*4 + ws*WM + wmo); enumerate q = wg*WS + ws
    float s = 0.f;
#pragma unroll 4
    for (int q = lane; q < nrec; q += 64) {      // (unrolled: the loads of a lane are independent, only the adds are ordered)
        const int wg = q / WS, w = q % WS;
        s += a.rec[((int64_t)wg * 4 + w * WM + wmo) * rec_floats + slot];
    }
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) {
        if (item == l0_count + head_count) { *loss_out = s * a.inv_count; return; }
        const int64_t e = item < l0_count ? item : off_head + (item - l0_count);
        a.grads[e] = s;
        if (a.update) {
            const float pv = optim_apply(opt, s, a.params, a.s1, a.s2, e);
            if (item < l0_count) {
                if (item < (int64_t)F * cin) a.pk[(item / cin) * 4 + (item % cin)] = brief_phase_scale(d, 0) * pv;     // W0p[o][c]
                else a.pk[(item - (int64_t)F * cin) * 4 + 3] = brief_phase_scale(d, 0) * pv;                           // W0p[o][3] = bias
            } else {
                const int64_t r = item - l0_count;
                float *hp = a.pk + brief_pk_head(d);
                if (r < (int64_t)cout * F) hp[(r / F) * FP + (r % F)] = pv;                   // Whp[c][f]
                else hp[4 * (int64_t)FP + (r - (int64_t)cout * F)] = pv;                      // bhp[c]
            }
        }
    }
}

__global__ __launch_bounds__(1024) void k_reduce(const ReduceArgs a, int nb_hidden)
{
    reduce_body(a, a.opt, a.loss_out, nb_hidden, (int)blockIdx.x);
}

// the reductions of co-trained jobs in one launch (see k_small_group): job j owns blocks [blk_begin[j], blk_begin[j + 1])
struct ReduceGroupArgs {
    const ReduceArgs *table;
    int njobs;
    int blk_begin[BRIEF_GROUP_MAX + 1];
    int nb_hidden[BRIEF_GROUP_MAX];
    OptimScalars opt[BRIEF_GROUP_MAX];
    float *loss_out[BRIEF_GROUP_MAX];
};
__global__ __launch_bounds__(1024) void k_reduce_group(const ReduceGroupArgs g)
{
    const int b = (int)blockIdx.x;
    const int j = group_job(g.blk_begin, g.njobs, b);
    ReduceArgs a;
    const_copy(a, g.table + j);      // (constant address space: see k_small_group)
    reduce_body(a, g.opt[j], g.loss_out[j], g.nb_hidden[j], b - g.blk_begin[j]);
}

#include "brief_bf16.inc"

// ---------------------------------------------------------------------------------------------
__global__ void k_repack(const brief_siren_desc d, const float *__restrict__ params, float *__restrict__ pk)
{
    const int F = d.features, cin = d.cin, cout = d.cout, L = d.layers;
    const int NT = brief_nt(d), FP = 32 * NT;
    const int64_t total = brief_pk_count(d);
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    float v = 0.f;
    const int64_t head = brief_pk_head(d);
    if (e >= brief_pk_count32(d)) {
        // bf16 fragment region (two elements per float slot); the alignment gap in front of it stays zero
        const int64_t base16 = brief_pk16_off(d, 1);
        uint32_t word = 0;
        if (e >= base16) {
            int64_t rr = e - base16;
            const bool lo_half = rr >= brief_pk16_region(d);      // BRIEF_PREC_BF16X3: the second region holds bf16(w - bf16(w))
            if (lo_half) rr -= brief_pk16_region(d);
            const int l = 1 + (int)(rr / ((int64_t)FP * FP));
            const float *W = params + brief_canon_hidden_off(d, l);
            for (int half = 0; half < 2; ++half) {
                int64_t q = (rr % ((int64_t)FP * FP)) * 2 + half;
                const bool bwd = q >= (int64_t)FP * FP;
                if (bwd) q -= (int64_t)FP * FP;
                const int j = (int)(q & 7), lanei = (int)((q >> 3) & 63), sstep = (int)((q >> 9) & 1);
                const int kt = (int)((q >> 10) % NT), mt = (int)((q >> 10) / NT);
                const int row = 32 * mt + (lanei & 31);
                const int col = 32 * kt + 16 * sstep + 8 * (j >> 2) + 4 * (lanei >> 5) + (j & 3);
                float w = 0.f;
                // the backward copy carries om_{l-1}: delta_{l-1} = (om W_l^T delta_l) . cos(om z_{l-1}); the forward copy om_l / 2 pi
                if (row < F && col < F) w = bwd ? brief_om_prev(d, l) * W[(int64_t)col * F + row] : brief_phase_scale(d, l) * W[(int64_t)row * F + col];
                uint16_t u16;
                if (d.precision == BRIEF_PREC_BF16X3) {
                    uint16_t hi16, lo16;
                    if (bwd) x3_split_bf16(w, hi16, lo16);
                    else x3_split_f16(BRIEF_X3_FWD_SCALE * w, hi16, lo16);
                    u16 = lo_half ? lo16 : hi16;
                } else {
                    union { __bf16 h; uint16_t u; } cv;
                    cv.h = (__bf16)w;
                    u16 = cv.u;
                }
                word |= (uint32_t)u16 << (16 * half);
            }
        }
        reinterpret_cast<uint32_t *>(pk)[e] = word;
        return;
    }
    if (e < (int64_t)FP * 4) {
        const int f = (int)(e >> 2), c = (int)(e & 3);
        if (f < F) {
            if (c < cin) v = brief_phase_scale(d, 0) * params[(int64_t)f * cin + c];
            else if (c == 3) v = brief_phase_scale(d, 0) * params[(int64_t)F * cin + f];
        }
    } else if (e < head) {
        const int64_t hs = brief_pk_hidden_stride(d);
        const int l = 1 + (int)((e - (int64_t)FP * 4) / hs);
        int64_t r = (e - (int64_t)FP * 4) % hs;
        const float *W = params + brief_canon_hidden_off(d, l);
        const float *b = W + (int64_t)F * F;
        if (r < 2 * (int64_t)FP * FP) {
            const bool bwd = r >= (int64_t)FP * FP;
            if (bwd) r -= (int64_t)FP * FP;
            const int jj = (int)(r & 3), lanei = (int)((r >> 2) & 63), q = (int)((r >> 8) & 3);
            const int kt = (int)((r >> 10) % NT), mt = (int)((r >> 10) / NT);
            const int row = 32 * mt + (lanei & 31);
            const int col = 32 * kt + 8 * q + 4 * (lanei >> 5) + jj;
            // (bf16 mode: the hidden GEMMs read the bf16 fragments only; these f32 slots stay zero, here and in k_reduce's write-through)
            if (row < F && col < F && d.precision != BRIEF_PREC_BF16) v = bwd ? brief_om_prev(d, l) * W[(int64_t)col * F + row] : brief_phase_scale(d, l) * W[(int64_t)row * F + col];
        } else {
            const int f = (int)(r - 2 * (int64_t)FP * FP);
            if (f < F) v = brief_phase_scale(d, l) * b[f];
        }
    } else {
        const int64_t r = e - head;
        const float *Wh = params + brief_canon_head_off(d);
        if (r < 4 * (int64_t)FP) {
            const int c = (int)(r / FP), f = (int)(r % FP);
            if (c < cout && f < F) v = Wh[(int64_t)c * F + f];
        } else {
            const int c = (int)(r - 4 * (int64_t)FP);
            if (c < cout) v = Wh[(int64_t)cout * F + c];
        }
    }
    (void)L;
    pk[e] = v;
}

__global__ void k_sample(int64_t *idx, int64_t n, uint64_t pop, uint64_t seed, uint64_t step)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = philox_index(i, pop, seed, step);
}

__global__ void k_sse_u16(const uint16_t *__restrict__ x, const uint16_t *__restrict__ y, int64_t n, unsigned long long *acc)
{
    unsigned long long s = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const long long dff = (long long)x[i] - (long long)y[i];
        s += (unsigned long long)(dff * dff);
    }
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(acc, s);   // integer: order-independent, exact
}

__global__ void k_u64_to_double(const unsigned long long *acc, double *out) { *out = (double)*acc; }

// ---------------------------------------------------------------------------------------------
// block-boundary ("deblock") filter of the reference: deblock.py:7-78 (mode 1, Python float arithmetic,
// pinned by goldens) / deblock.cpp:12-71,277-319 (mode 0, C integer arithmetic).  One launch filters one
// boundary line for every slice z1..z2; a thread owns one position along the line and the 6 pixels
// across it, so threads of a launch touch disjoint pixels.  Lines are launched in the reference's order
// (the filter is in place and neighbouring lines overlap), see brief_pytorch_amd/deblock.py.
__global__ void k_deblock_edge(uint16_t *img, int64_t H, int64_t W, int z1, int nz, int fixed, int a1, int na, int vertical,
                               double alpha, double beta, double thres, int mode)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)nz * na) return;
    const int z = z1 + (int)(t / na), pos = a1 + (int)(t % na);
    const int64_t stride = vertical ? 1 : W;
    uint16_t *p = img + (int64_t)z * H * W + (vertical ? (int64_t)pos * W + fixed : (int64_t)fixed * W + pos);
    const int P2 = p[-3 * stride], P1 = p[-2 * stride], P0 = p[-stride], Q0 = p[0], Q1 = p[stride], Q2 = p[2 * stride];
    if (mode == 1) {
        const double p2 = P2, q2 = Q2;
        double p1 = P1, p0 = P0, q0 = Q0, q1 = Q1;
        if ((p1 + p0 + q0 + q1) / 4 > thres) return;
        if (!(fabs(p0 - q0) < alpha && fabs(p1 - p0) < beta && fabs(q1 - q0) < beta)) return;
        double delta0 = (4 * (q0 - p0) + (p1 - q1) + 4) / 8;
        double deltap1 = (p2 + (p0 + q0 + 1) / 2 - 2 * p1) / 2;
        double deltaq1 = (q2 + (q0 + p0 + 1) / 2 - 2 * q1) / 2;
        double c1 = 20, c0 = c1;
        if (fabs(p2 - p0) < beta) c0 += 1;
        if (fabs(q2 - q0) < beta) c0 += 1;
        delta0 = fmin(fmax(delta0, -c0), c0);
        deltap1 = fmin(fmax(deltap1, -c1), c1);
        deltaq1 = fmin(fmax(deltaq1, -c1), c1);
        p1 += deltap1; p0 += delta0; q0 -= delta0; q1 += deltaq1;
        p[-2 * stride] = (uint16_t)(long long)p1; p[-stride] = (uint16_t)(long long)p0;
        p[0] = (uint16_t)(long long)q0; p[stride] = (uint16_t)(long long)q1;
    } else {
        if ((P1 + P0 + Q0 + Q1) / 4 > (int)thres) return;
        const float al = (float)alpha, be = (float)beta;
        if (!((float)abs(P0 - Q0) < al && (float)abs(P1 - P0) < be && (float)abs(Q1 - Q0) < be)) return;
        float delta0 = (float)((4 * (Q0 - P0) + (P1 - Q1) + 4) / 8);
        float deltap1 = (float)((P2 + (P0 + Q0 + 1) / 2 - 2 * P1) / 2);
        float deltaq1 = (float)((Q2 + (Q0 + P0 + 1) / 2 - 2 * Q1) / 2);
        float c1 = 20.f, c0 = 20.f;
        if ((float)abs(P2 - P0) < be) c0 += 1.f;
        if ((float)abs(Q2 - Q0) < be) c0 += 1.f;
        delta0 = fminf(fmaxf(delta0, -c0), c0);
        deltap1 = fminf(fmaxf(deltap1, -c1), c1);
        deltaq1 = fminf(fmaxf(deltaq1, -c1), c1);
        p[-2 * stride] = (uint16_t)(int)((float)P1 + deltap1); p[-stride] = (uint16_t)(int)((float)P0 + delta0);
        p[0] = (uint16_t)(int)((float)Q0 - delta0); p[stride] = (uint16_t)(int)((float)Q1 + deltaq1);
    }
}

// ---------------------------------------------------------------------------------------------
// SSIM of the reference (utils/ssim.py:9-150 as called by utils/misc.py:458-475): per z-slice 2-D SSIM with an
// 11-tap sigma-1.5 separable Gaussian (valid padding, H pass then W pass, f32), K = (0.01, 0.03); the caller
// averages the per-slice means.  One block = one 16x64 output tile of one slice; partial sums in double.
#define SSIM_TH 16
#define SSIM_TW 64
__global__ __launch_bounds__(256) void k_ssim_u16(const uint16_t *__restrict__ X, const uint16_t *__restrict__ Y, int64_t H, int64_t W,
                                                    int tiles_h, int tiles_w, const float *__restrict__ win, float C1, float C2,
                                                    double *__restrict__ partial)
{
    __shared__ float sx[SSIM_TH + 10][SSIM_TW + 10], sy[SSIM_TH + 10][SSIM_TW + 10];
    __shared__ float v[5][SSIM_TH][SSIM_TW + 10];
    __shared__ float w[11];
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int64_t z = blockIdx.x / (tiles_h * tiles_w);
    const int th = (blockIdx.x / tiles_w) % tiles_h, tw = blockIdx.x % tiles_w;
    const int64_t oh = H - 10, ow = W - 10;                 // valid output extent
    const int64_t r0 = (int64_t)th * SSIM_TH, c0 = (int64_t)tw * SSIM_TW;
    if (tid < 11) w[tid] = win[tid];
    const uint16_t *xs = X + z * H * W, *ys = Y + z * H * W;
    for (int e = tid; e < (SSIM_TH + 10) * (SSIM_TW + 10); e += 256) {
        const int r = e / (SSIM_TW + 10), c = e % (SSIM_TW + 10);
        const int64_t gr = r0 + r, gc = c0 + c;
        float a = 0.f, b = 0.f;
        if (gr < H && gc < W) { a = (float)xs[gr * W + gc]; b = (float)ys[gr * W + gc]; }
        sx[r][c] = a; sy[r][c] = b;
    }
    __syncthreads();
    for (int e = tid; e < SSIM_TH * (SSIM_TW + 10); e += 256) {      // H pass
        const int r = e / (SSIM_TW + 10), c = e % (SSIM_TW + 10);
        float m1 = 0.f, m2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float a = sx[r + k][c], b = sy[r + k][c], wk = w[k];
            m1 += wk * a; m2 += wk * b; s11 += wk * (a * a); s22 += wk * (b * b); s12 += wk * (a * b);
        }
        v[0][r][c] = m1; v[1][r][c] = m2; v[2][r][c] = s11; v[3][r][c] = s22; v[4][r][c] = s12;
    }
    __syncthreads();
    double acc = 0.0;
    for (int e = tid; e < SSIM_TH * SSIM_TW; e += 256) {             // W pass + SSIM map
        const int r = e / SSIM_TW, c = e % SSIM_TW;
        if (r0 + r < oh && c0 + c < ow) {
            float m1 = 0.f, m2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float wk = w[k];
                m1 += wk * v[0][r][c + k]; m2 += wk * v[1][r][c + k]; s11 += wk * v[2][r][c + k];
                s22 += wk * v[3][r][c + k]; s12 += wk * v[4][r][c + k];
            }
            const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
            const float sg1 = s11 - m11, sg2 = s22 - m22, sg12 = s12 - m12;
            const float cs = (2.f * sg12 + C2) / (sg1 + sg2 + C2);
            acc += (double)(((2.f * m12 + C1) / (m11 + m22 + C1)) * cs);
        }
    }
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

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
        return fail(BRIEF_ERR_INVALID, "features must be 1..1024 on the fused path");
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
static hipEvent_t g_prof_ev[2 * kProfSlots];
static bool g_prof_init = false;

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
static const int g_reduce_sg_small = env_int("BRIEF_REDUCE_SG", 0, 0, 64);      // diagnostics: k_reduce threads per hidden parameter behind k_small (power of two; 0 = by slab count)
// samples one workgroup tile covers: 32 per sample sub-tile; the split-precision TRAIN kernel walks 64-sample tiles
static int64_t fused_wg_samples(const brief_siren_desc &d, bool train)
{
    if (d.precision == BRIEF_PREC_F32 && brief_use_lean(brief_nt(d), train)) return 32;              // k_lean<1, ...>: one 32-sample tile
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
    if (d.precision == BRIEF_PREC_F32 && brief_use_lean(nt, train))      // k_lean: what its launch bounds and its LDS image allow
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
    if (!train || nt > 8 || tiles <= cap) return p;        // decode: millions of tiles, the tail does not matter; > 8 tiles: one workgroup per CU
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
static int wgrad_splits(const brief_siren_desc &d, int64_t n)
{
    const int hidden = d.layers - 2;
    if (hidden <= 0) return 0;
    const int64_t nchunks = brief_npad_d(d, n) / 32;
    int64_t s = kWgradBlocks / (hidden * wgrad_nq(brief_nt(d)) * wgrad_nq(brief_nt(d)));
    if (s < 1) s = 1;
    if (s > nchunks) s = nchunks;
    return (int)s;
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
static int wgrad16_splits(const brief_siren_desc &d, int64_t n)
{
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

struct WsLayout { int64_t z, dd, rec, slabs, table, total; };
static WsLayout ws_layout(const brief_siren_desc &d, int64_t n)
{
    if (d.precision == BRIEF_PREC_BF16) {
        WsLayout w; w.z = w.dd = w.rec = w.slabs = w.table = 0; w.total = ws16_layout(d, n).total;
        return w;
    }
    const int nt = brief_nt(d);
    const int64_t FP = 32 * nt, npad = brief_npad_d(d, n), hidden = d.layers - 2 > 0 ? d.layers - 2 : 0;
    const bool small = use_small(d);
    WsLayout w;
    w.z = 0;
    w.dd = w.z + (small ? 0 : hidden * FP * npad);
    w.rec = w.dd + (small ? 0 : hidden * FP * npad);
    w.slabs = w.rec + (int64_t)kCUs * kRecWgsPerCu * 4 * brief_rec_floats(nt);
    w.table = w.slabs + hidden * (int64_t)(small ? small_grid(d, n) : wgrad_splits(d, n)) * (FP * FP + FP);
    w.table = (w.table + 3) / 4 * 4;
    w.total = w.table + (small ? kGroupTableFloats : 0);
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
    if (e0) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), (uint32_t)lds, st, e0, e1, 0, arg);
    else hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, st, arg);
}

template <bool TRAIN>
static int launch_fused(const FusedArgs &fa, int grid, hipStream_t st, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr)
{
    const int nt = brief_nt(fa.d);
    if (fa.d.precision == BRIEF_PREC_BF16X3 && (TRAIN || g_x3_decode)) {
        // split precision: the 64-sample walk (k_fused_x3); BRIEF_X3_DECODE=0 evaluates such a net on the f32 forward kernel below
        const size_t lds = sizeof(float) * X3TLds::TOTAL;
        static bool attr_t64 = false;
        if (!attr_t64) {
            HIP_TRY(hipFuncSetAttribute((const void *)k_fused_x3<TRAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_t64 = true;
        }
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
        static bool attr_64 = false;
        if (!attr_64) {
            HIP_TRY(hipFuncSetAttribute((const void *)k_lean<SHV, 2, 8, TRAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_64 = true;
        }
        launch_timed(k_lean<SHV, 2, 8, TRAIN>, grid, 256, lds, st, fa, e0, e1);
        HIP_TRY(hipGetLastError());
        return 0;
    }
#endif
    if (brief_use_lean(nt, TRAIN)) {
        // k_lean<1, MTW, 0>: a run-time number of feature tiles, MTW = ceil(nt / 4) of them per wave (brief_layout.h: brief_use_lean says
        // which widths; inference of exactly 12 or 16 tiles stays on k_fused<12 / 16, false>, whose unrolled chains decode 5 % faster:
        // 4x384 0.89 against 0.84 of the fp32 peak, 4x512 0.92 against 0.87 — tools/decode_widths.py; the layouts are the same)
        const int mtw = (nt + 3) / 4;
        const int rm = BRIEF_LEAN_RM3 ? (nt & 3) : ((nt & 3) == 3 ? 0 : (nt & 3));      // left-over tiles shared along K by the four waves (brief_lean.inc)
        const size_t lds = sizeof(float) * lean_lds(1, mtw, nt).total;
        static bool attr_w[9][4] = {};
#define BRIEF_WIDE_RM(MTWV, RMV)                                                                         \
    {                                                                                                    \
        if (!attr_w[MTWV][RMV]) {                                                                        \
            HIP_TRY(hipFuncSetAttribute((const void *)k_lean<1, MTWV, 0, TRAIN, RMV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)(sizeof(float) * lean_lds(1, MTWV, 4 * MTWV).total)));      \
            attr_w[MTWV][RMV] = true;                                                                    \
        }                                                                                                \
        launch_timed(k_lean<1, MTWV, 0, TRAIN, RMV>, grid, 256, lds, st, fa, e0, e1);                    \
    }
#define BRIEF_WIDE(MTWV)                                                                                 \
    case MTWV:                                                                                           \
        if (rm == 1) BRIEF_WIDE_RM(MTWV, 1) else if (rm == 2) BRIEF_WIDE_RM(MTWV, 2) else if (rm == 3) BRIEF_WIDE_RM(MTWV, 3) else BRIEF_WIDE_RM(MTWV, 0)   \
        break;
        switch (mtw) {
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
    case 7:
    case 12:
    case 16:
        if constexpr (!TRAIN) {      // inference only (their TRAIN steps run on k_lean)
            const size_t lds7 = sizeof(float) * FusedLds<7, true>::TOTAL, lds12 = sizeof(float) * FusedLds<12, true>::TOTAL, lds16 = sizeof(float) * FusedLds<16, true>::TOTAL;
            if (nt == 7) launch_timed(k_fused<7, false>, grid, 256, lds7, st, fa, e0, e1);
            else if (nt == 12) launch_timed(k_fused<12, false>, grid, 256, lds12, st, fa, e0, e1);
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
    static bool attr_done[2][2][2][3] = {};
    bool launched = false;
#define BRIEF_CASE(NTV, COV, NSV)                                                                        \
    if (!launched && nt == NTV && (fa.d.cout == 1) == (COV == 1) && ns == NSV) {                         \
        const size_t lds = sizeof(float) * Cfg16<NTV, NSV>::TOTAL;                                       \
        bool &done = attr_done[NTV == 16][TRAIN][COV == 1][NSV == 4 ? 2 : NSV - 1];                      \
        if (!done) {                                                                                     \
            HIP_TRY(hipFuncSetAttribute((const void *)k16<NTV, TRAIN, COV, NSV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            done = true;                                                                                 \
        }                                                                                                \
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

int brief_siren_forward(const brief_siren_desc *d, const float *packed, const brief_grid_desc *grid,
                        const brief_batch_desc *batch, void *out, int out_kind,
                        float scale_min, float scale_max, double vmin, double vmax, void *stream)
{
    if (int rc = check_desc(d)) return rc;
    if (int rc = check_batch(d, grid, batch, false)) return rc;
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
    if (d->precision == BRIEF_PREC_BF16) return launch_k16_split<false>(fa, (hipStream_t)stream);
    const FusedPlan fp = fused_plan(*d, batch->n, false);
    fa.pers_wgs = fp.pers_wgs; fa.pers_tiles = fp.pers_tiles;
    return launch_fused<false>(fa, fp.grid, (hipStream_t)stream);
}

struct UpdatePayload { OptimScalars opt; float *params, *s1, *s2, *pk; };

}   // extern "C"

// what one narrow-net step launches, as data: brief_multi_fit collects these for a group of jobs and launches them together
struct SmallStepPlan { FusedArgs fa; ReduceArgs ra; int grid1, nb_hidden, nb_reduce, nt, hb; };

static int train_step_impl(const brief_siren_desc *d, const float *packed, const brief_grid_desc *grid,
                           const brief_batch_desc *batch, int loss_kind, float thr, float beta,
                           float *grads, float *loss_out, float *yhat_out,
                           void *workspace, int64_t workspace_bytes, void *stream, const UpdatePayload *upd, SmallStepPlan *plan_only = nullptr)
{
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
        const bool prof16 = g_prof_on && g_prof_n < kProfSlots;
        if (prof16) HIP_TRY(hipEventRecord(g_prof_ev[2 * g_prof_n], st));
        if (int rc = launch_k16_split<true>(fa, st)) return rc;
        if (prof16) { HIP_TRY(hipEventRecord(g_prof_ev[2 * g_prof_n + 1], st)); ++g_prof_n; }
        Wgrad16Args wa;
        memset(&wa, 0, sizeof(wa));
        const int nsp_s = wgrad16_skinny_splits(*d, batch->n);
        wa.d = *d; wa.npad = np; wa.nsplit = nsp; wa.nsplit_s = nsp_s; wa.bias_jobs = 0; wa.slabs = ws + w16.slabs;   // (bias_jobs: k_wgrad16_big computes db_l itself since round 2)
        wa.H = (const __bf16 *)(ws + w16.h); wa.D = (const __bf16 *)(ws + w16.dd);
        wa.X = (const __bf16 *)(ws + w16.x); wa.G = (const __bf16 *)(ws + w16.g);
        const int nb = nt / 4;
        if (hidden > 0 && nt == 16) {
            static bool big_attr = false;
            const int lds_big = (int)(sizeof(float) * 4 * W16B_PANEL);
            if (!big_attr) { HIP_TRY(hipFuncSetAttribute((const void *)k_wgrad16_big, hipFuncAttributeMaxDynamicSharedMemorySize, lds_big)); big_attr = true; }
            const int groups8 = (hidden * nsp + 7) / 8 * 8;        // (layer, split) groups padded to whole XCD rounds (see the kernel)
            hipLaunchKernelGGL(k_wgrad16_big, dim3(groups8 * 4), dim3(512), lds_big, st, wa);
        } else if (hidden > 0) {
            hipLaunchKernelGGL(k_wgrad16<false>, dim3(hidden * nb * nb * nsp), dim3(256), sizeof(float) * 4 * W16_PANEL, st, wa);
        }
        // hidden-layer bias gradients (B = ones), first layer, head
        hipLaunchKernelGGL(k_wgrad16<true>, dim3((wa.bias_jobs ? hidden * nb * nsp : 0) + 2 * nb * nsp_s), dim3(256), sizeof(float) * 4 * W16_PANEL, st, wa);
        HIP_TRY(hipGetLastError());
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
    const int grid1 = small ? small_grid(*d, batch->n) : fp.grid;
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
    fa.rec = ws + wl.rec; fa.slabs = ws + wl.slabs; fa.yhat_out = yhat_out;
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
    const int64_t skinny = l0_count + (int64_t)d->cout * d->features + d->cout + 1;   // + the loss
    const int ppb = rthreads / ra.sgroups;             // hidden parameters per k_reduce block
    const int nb_hidden = (int)((hcount + ppb - 1) / ppb);
    const int wpb = rthreads / 64;                     // skinny items (one wave each) per block
    const int nb_skinny = (int)((skinny + wpb - 1) / wpb);
    if (plan_only) {
        if (!small) return fail(BRIEF_ERR_INVALID, "internal: step plans exist for narrow nets only");
        plan_only->fa = fa; plan_only->ra = ra; plan_only->grid1 = grid1; plan_only->nb_hidden = nb_hidden;
        plan_only->nb_reduce = nb_hidden + nb_skinny; plan_only->nt = nt; plan_only->hb = small_hb(*d);
        return 0;
    }
    const bool prof = g_prof_on && g_prof_n < kProfSlots;
    hipEvent_t pe0 = prof ? g_prof_ev[2 * g_prof_n] : nullptr, pe1 = prof ? g_prof_ev[2 * g_prof_n + 1] : nullptr;
    if (small) {
        const int hb = small_hb(*d);
#define BRIEF_CASE(NTV, HBV)                                                                                        \
    if (nt == NTV && hb == HBV)                                                                                     \
        launch_timed(k_small<NTV, HBV>, grid1, 256, sizeof(float) * SmallLds<NTV>::TOTAL, st, fa, pe0, pe1);
        BRIEF_CASE(1, 1) BRIEF_CASE(1, 3) BRIEF_CASE(1, 5) BRIEF_CASE(1, 7)
        BRIEF_CASE(2, 1) BRIEF_CASE(2, 3) BRIEF_CASE(2, 5) BRIEF_CASE(2, 7)
#undef BRIEF_CASE
        HIP_TRY(hipGetLastError());
    } else if (int rc = launch_fused<true>(fa, grid1, st, pe0, pe1)) return rc;
    if (prof) ++g_prof_n;

    if (!small && nsplit > 0) {
        WgradArgs wa;
        memset(&wa, 0, sizeof(wa));
        wa.d = *d; wa.Z = fa.Z; wa.D = fa.D; wa.npad = fa.npad; wa.nsplit = nsplit; wa.slabs = ws + wl.slabs;
        wa.stamps = ws + wl.rec + (int64_t)kCUs * kRecWgsPerCu * 4 * brief_rec_floats(nt) - 256 * 8 * 8;   // tail of the record region (diagnostics)
        const int blocks = nsplit * (d->layers - 2) * wgrad_nq(nt) * wgrad_nq(nt);
        if (d->precision == BRIEF_PREC_BF16X3) {
            for (int rep = 0; rep < g_wgrad_repeat; ++rep)      // BRIEF_WGRAD_REPEAT (diagnostics): the launch is idempotent
                hipLaunchKernelGGL(k_wgrad_x3, dim3(blocks), dim3(512), sizeof(float) * wgrad_lds_floats(8), st, wa);
        } else
#define BRIEF_CASE(NTV)                                                                                    \
    case NTV:                                                                                              \
        hipLaunchKernelGGL((k_wgrad<NTV>), dim3(blocks), dim3(512), sizeof(float) * wgrad_lds_floats(NTV), st, wa); \
        break;
        if (nt > 8) {
            // run-time width: ceil(nt / 8) quadrants per side of QT = ceil(nt / quadrants) tiles (the last row / column may be short)
            const int qt = (nt + wgrad_nq(nt) - 1) / wgrad_nq(nt);
            switch (qt) {
            case 5: hipLaunchKernelGGL((k_wgrad<0, 5>), dim3(blocks), dim3(512), sizeof(float) * wgrad_lds_floats(5), st, wa); break;
            case 6: hipLaunchKernelGGL((k_wgrad<0, 6>), dim3(blocks), dim3(512), sizeof(float) * wgrad_lds_floats(6), st, wa); break;
            case 7: hipLaunchKernelGGL((k_wgrad<0, 7>), dim3(blocks), dim3(512), sizeof(float) * wgrad_lds_floats(7), st, wa); break;
            case 8: hipLaunchKernelGGL((k_wgrad<0, 8>), dim3(blocks), dim3(512), sizeof(float) * wgrad_lds_floats(8), st, wa); break;
            default: return fail(BRIEF_ERR_INVALID, "unsupported width");
            }
        } else
        switch (nt) {
            BRIEF_CASE(1) BRIEF_CASE(2) BRIEF_CASE(3) BRIEF_CASE(4)
            BRIEF_CASE(5) BRIEF_CASE(6) BRIEF_CASE(7) BRIEF_CASE(8)
        }
#undef BRIEF_CASE
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(k_reduce, dim3(nb_hidden + nb_skinny), dim3(rthreads), 0, st, ra, nb_hidden);
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

static const int kPoolStreams = 8;
static hipStream_t g_pool[kPoolStreams];
static hipEvent_t g_pool_ev[kPoolStreams + 1];
static bool g_pool_init = false;

int brief_multi_fit(const brief_fit_job *jobs, int32_t njobs, int64_t steps, void *stream)
{
    if (njobs < 1 || !jobs) return fail(BRIEF_ERR_INVALID, "no jobs");
    if (steps < 0) return fail(BRIEF_ERR_INVALID, "bad step count");
    if (njobs > 4096) return fail(BRIEF_ERR_INVALID, "too many jobs for one call");
    for (int j = 0; j < njobs; ++j)
        if (int rc = fit_job_check(&jobs[j])) return rc;
    if (njobs == 1) return brief_siren_fit(jobs, steps, stream);
    if (!g_pool_init) {
        for (int s = 0; s < kPoolStreams; ++s) HIP_TRY(hipStreamCreateWithFlags(&g_pool[s], hipStreamNonBlocking));
        for (int s = 0; s <= kPoolStreams; ++s) HIP_TRY(hipEventCreateWithFlags(&g_pool_ev[s], hipEventDisableTiming));
        g_pool_init = true;
    }
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
        if (!use_small(jobs[j].desc) || (g_prof_on && g_prof_n < kProfSlots)) continue;      // (timed runs keep their per-job launches)
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
    if (on && !g_prof_init) {
        for (int i = 0; i < 2 * kProfSlots; ++i) HIP_TRY(hipEventCreate(&g_prof_ev[i]));
        g_prof_init = true;
    }
    g_prof_on = on != 0;
    g_prof_n = 0;
    return 0;
}

int brief_profile_fused(double *total_ms, int64_t *launches)
{
    if (!total_ms || !launches) return fail(BRIEF_ERR_INVALID, "null output");
    double tot = 0.0;
    for (int i = 0; i < g_prof_n; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(g_prof_ev[2 * i + 1]));
        HIP_TRY(hipEventElapsedTime(&ms, g_prof_ev[2 * i], g_prof_ev[2 * i + 1]));
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
