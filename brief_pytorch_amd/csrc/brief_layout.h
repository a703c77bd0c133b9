// brief_layout.h — buffer layouts shared by the kernels and the host-side launch code.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../../include/brief_hip.h"

#if defined(__HIPCC__)
#define BL_HD __host__ __device__ __forceinline__
#else
#define BL_HD static inline
#endif

#define BRIEF_MAX_NT 128        // features <= 4096 (byte offsets inside one weight block / stash plane are 32-bit); the width is padded to whole 32-feature tiles only
// per-wave partial record of the fused kernels: dW0[TR local features][4] | dWh[4][TR] | dbh[4] | loss | stamps, TR = 128 local
// features per wave (4 feature tiles) up to 512 features, 256 (8 tiles) up to 1024, 32 ceil(nt / 4) above (k_wide): brief_rec_tr.  The constants are the TR = 128 case.
#define BRIEF_REC_FLOATS 1056
#define BRIEF_REC_DWH 512
#define BRIEF_REC_DBH 1024
#define BRIEF_REC_LOSS 1028
#define BRIEF_REC_STAMPS 1030
BL_HD int brief_rec_tr(int nt) { return nt > 32 ? 32 * ((nt + 3) / 4) : (nt > 16 ? 256 : 128); }
BL_HD int brief_rec_floats(int nt) { return 8 * brief_rec_tr(nt) + 32; }

// --- canonical parameter buffer: W0[F,cin] b0[F] | (W_l[F,F] b_l[F]) x (L-2) | Wh[cout,F] bh[cout]
BL_HD int64_t brief_canon_hidden_off(const brief_siren_desc &d, int l /*1..L-2*/)
{
    const int64_t F = d.features;
    return F * d.cin + F + (int64_t)(l - 1) * (F * F + F);
}
BL_HD int64_t brief_canon_head_off(const brief_siren_desc &d)
{
    const int64_t F = d.features;
    return F * d.cin + F + (int64_t)(d.layers - 2) * (F * F + F);
}
BL_HD int64_t brief_canon_count(const brief_siren_desc &d)
{
    return brief_canon_head_off(d) + (int64_t)d.features * d.cout + d.cout;
}

// --- derived ("packed") buffer, all rows padded to FP = 32*NT with zeros.  The copies carry the sine frequencies, so that
//     the matrix pipe delivers what the epilogues need and no VALU instruction is spent on it (v_mfma_f32_32x32x2_f32 and
//     the VALU share the SIMD's FMA lanes: profiles/r03_coissue.md):  s_l = w0_l / 2 pi  (brief_phase_scale) turns
//     z_l into the PHASE of the following sine in revolutions, which v_sin_f32 / v_cos_f32 take as they are.
//   W0p  [FP][4]            = s_0 * {w(x0), w(x1), w(x2) (0 if cin==2), bias}
//   per hidden layer l=1..L-2:
//     Wf [NT(mt)][NT(kt)][4(q)][64(lane)][4(j)]  A-fragments of s_l W       : W[32mt+i][32kt+8q+4hi+j]
//     Wb [NT(mt)][NT(kt)][4(q)][64(lane)][4(j)]  A-fragments of w0_{l-1} W^T : W[32kt+8q+4hi+j][32mt+i]
//                                                 (delta_{l-1} = (w0_{l-1} W_l^T delta_l) . cos(phase_{l-1}))
//     bp [FP]                                     s_l * bias
//   Whp [4][FP]  (rows >= cout zero),  bhp[4]     (unscaled)
//   (lane = 32*hi + i ; this is the operand order of v_mfma_f32_32x32x2_f32, see brief_device.inc)
// number of 32-feature tiles of the width: exact for every fp32 net (1 .. 32 tiles)
BL_HD int brief_nt(const brief_siren_desc &d)
{
    const int nt = (d.features + 31) / 32;
    if (d.precision == BRIEF_PREC_BF16) return nt <= 8 ? 8 : 16;     // the bf16 kernels exist for 256 and 512 padded features
    if (d.precision == BRIEF_PREC_BF16X3) return 8;                  // split precision: one kernel set, 256 padded features (check_desc: F <= 256)
    return nt;
}
// Which exact-f32 kernel walks a net of nt tiles (round 4; measured step / decode fractions in profiles/r04_widths.md):
//   k_fused<NT>  compile-time width, fully unrolled chains, four register arrays per wave: 1, 2, 4 tiles and the 8-tile headline (TRAIN);
//                1 .. 4, 8, 12, 16 tiles (inference: its unrolled chains decode the exact 12- / 16-tile widths 5 % faster).
//   k_lean       run-time width, rolled chains, one register array per wave, left-over tiles (nt % 4 = 1, 2, 3) shared along K: everything
//                else — 3 tiles (TRAIN: each of the four waves carries 3 / 4 of a tile pass where k_fused<3>'s fourth wave idled: 4x96 kernel 0.165 -> 0.154 ms,
//                4x65 0.145 -> 0.136), 5 .. 7 tiles (4x160 trains at 0.61 of the fp32 peak against 0.52, 4x192 0.67 against 0.64) and 9 .. 32 tiles.
#ifndef BRIEF_LEAN3
#define BRIEF_LEAN3 1      // 3-tile nets (65 .. 96 features) TRAIN on k_lean, their three tiles shared along K by the four waves (0: k_fused<3>, whose fourth wave idles)
#endif
//   k_wide       33 .. 128 tiles (1025 .. 4096 features): k_lean's rolled chain on K-slabs staged from the stash planes, output tiles in passes (brief_wide.inc)
BL_HD bool brief_use_wide(const brief_siren_desc &d) { return d.precision == BRIEF_PREC_F32 && (d.features + 31) / 32 > 32; }
BL_HD bool brief_use_lean(const brief_siren_desc &d, bool train)
{
    const int nt = brief_nt(d);
    if (nt > 32) return false;
    if (train) return (nt >= 5 && nt != 8) || (BRIEF_LEAN3 && nt == 3);
    // (inference, 7 tiles: 4x224 decodes a 256^3 grid in 39.1 ms against k_fused<7, false>'s 42.5, 4x200 in 36.3 against 38.7 since the rolled chain stops at ceil(F / 8) steps)
    return (nt >= 5 && nt <= 7) || (nt >= 9 && nt != 12 && nt != 16);
}
BL_HD int64_t brief_pk_w0(const brief_siren_desc &) { return 0; }
BL_HD int64_t brief_pk_hidden_stride(const brief_siren_desc &d)
{
    const int64_t FP = 32 * brief_nt(d);
    return 2 * FP * FP + FP;
}
BL_HD int64_t brief_pk_hidden(const brief_siren_desc &d, int l /*1..L-2*/)
{
    const int64_t FP = 32 * brief_nt(d);
    return FP * 4 + (int64_t)(l - 1) * brief_pk_hidden_stride(d);
}
BL_HD int64_t brief_pk_head(const brief_siren_desc &d)
{
    const int64_t FP = 32 * brief_nt(d);
    return FP * 4 + (int64_t)(d.layers - 2) * brief_pk_hidden_stride(d);
}
BL_HD int64_t brief_pk_count32(const brief_siren_desc &d)
{
    const int64_t FP = 32 * brief_nt(d);
    return brief_pk_head(d) + 4 * FP + 4;
}
// bf16 mode appends, per hidden layer, the bf16 A-fragments of W and of W^T for v_mfma_f32_32x32x16_bf16:
//   Wf16 [NT(mt)][NT(kt)][2(s)][64(lane)][8(j)]:  s_l W[32mt + i][32kt + krow(s, hi, j)]       (lane = 32 hi + i)
//   Wb16 [NT(mt)][NT(kt)][2(s)][64(lane)][8(j)]:  w0_{l-1} W[32kt + krow(s, hi, j)][32mt + i]
//   krow(s, hi, j) = 16 s + 8 (j >> 2) + 4 hi + (j & 3): the row an accumulator register 8s + j of lane half hi
//   holds, so that a converted accumulator tile IS the B operand of the next layer (cdna_hip_programming.md,
//   'An accumulator tile as the next MFMA's operand').  Offsets below are in FLOAT units (2 bf16 each).
BL_HD int64_t brief_pk16_off(const brief_siren_desc &d, int l /*1..L-2*/)
{
    const int64_t FP = 32 * brief_nt(d);
    return ((brief_pk_count32(d) + 3) / 4) * 4 + (int64_t)(l - 1) * FP * FP;
}
// BRIEF_PREC_BF16X3 keeps the whole f32 buffer (the decode kernels read it) and appends TWO bf16 regions of the bf16 mode's
// shape: the hi halves bf16(w) and, (L-2) FP^2 floats further, the lo halves bf16(w - hi) of the same (scaled) weights
BL_HD int64_t brief_pk16_region(const brief_siren_desc &d)
{
    const int64_t FP = 32 * brief_nt(d), hidden = d.layers - 2 > 0 ? d.layers - 2 : 0;
    return hidden * FP * FP;
}
BL_HD int64_t brief_pk_count(const brief_siren_desc &d)
{
    if (d.precision == BRIEF_PREC_BF16) return brief_pk16_off(d, 1) + brief_pk16_region(d);
    if (d.precision == BRIEF_PREC_BF16X3) return brief_pk16_off(d, 1) + 2 * brief_pk16_region(d);
    return brief_pk_count32(d);
}
// index (in bf16 elements, inside one layer's Wf16 or Wb16 block) of the fragment element that multiplies
// activation row `col` into output row `row`
BL_HD int64_t brief_frag16_index(int NT, int row, int col)
{
    const int mt = row >> 5, i = row & 31, kt = col >> 5, rem = col & 31;
    const int s = rem >> 4, r16 = rem & 15, hi = (r16 >> 2) & 1, j = ((r16 >> 3) << 2) | (r16 & 3);
    return ((((int64_t)mt * NT + kt) * 2 + s) * 64 + 32 * hi + i) * 8 + j;
}

// w0 of the sine layer below hidden layer l (1..L-2): the factor folded into the copies of W_l^T (f32 and bf16)
BL_HD float brief_om_prev(const brief_siren_desc &d, int l) { return l - 1 == 0 ? d.w0_first : d.w0_hidden; }
// w0_l / 2 pi: the factor folded into layer l's forward weights and bias (l = 0: first layer, l >= 1: hidden layers)
BL_HD float brief_phase_scale(const brief_siren_desc &d, int l) { return (l == 0 ? d.w0_first : d.w0_hidden) * 0.15915494309189535f; }

// --- fused-kernel geometry per NT (how the 4 waves of a workgroup split features x sample tiles)
// Five and six feature tiles: two waves per sample tile x three tile slots each, two sample tiles per workgroup (with four waves x
// two slots, 3 / 2 of the 8 slots were empty: 5x192 step 0.769 -> 0.701 ms, 5x160 0.598 -> 0.584); 7 tiles keep the 4 x 2 slots.
BL_HD constexpr int brief_wm(int nt) { return nt >= 3 ? 4 : nt; }          // waves along features (TRAIN kernels, k_reduce; 5 and 6 tiles were 2 x 3 slots on k_fused until round 4)
// ... in the inference kernels (no records for k_reduce to agree with): three tiles -> one wave per sample tile owns all three (no idle
// fourth wave: 256^3 decode of a 4x96 net 11.1 -> 8.7 ms; the TRAIN kernel spills in that form); five tiles keep 4 x 2 (25.1 ms against
// 27.1 ms for a 4x160 net); six as above (34.2 -> 29.9 ms)
BL_HD constexpr int brief_wm_infer(int nt) { return nt == 3 ? 1 : (nt == 5 ? 4 : brief_wm(nt)); }
BL_HD constexpr int brief_ws(int nt) { return 4 / brief_wm(nt); }          // sample tiles (of 32) per workgroup
BL_HD int64_t brief_wg_samples(int nt) { return 32 * brief_ws(nt); }
BL_HD int64_t brief_npad(int nt, int64_t n)
{
    const int64_t g = brief_wg_samples(nt);
    return (n + g - 1) / g * g;
}
// padded sample count of the TRAIN stash planes
BL_HD int64_t brief_npad_d(const brief_siren_desc &d, int64_t n) { return brief_npad(brief_nt(d), n); }
