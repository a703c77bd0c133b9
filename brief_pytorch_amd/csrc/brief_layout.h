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

#define BRIEF_MAX_NT 16         // features <= 512 (widths above 256 are padded to 384 or 512)
// per-wave partial record of the fused kernel: dW0[128 local features][4] | dWh[4][128] | dbh[4] | loss | stamps
#define BRIEF_REC_FLOATS 1056
#define BRIEF_REC_DWH 512
#define BRIEF_REC_DBH 1024
#define BRIEF_REC_LOSS 1028
#define BRIEF_REC_STAMPS 1030

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

// --- derived ("packed") buffer, all rows padded to FP = 32*NT with zeros:
//   W0p  [FP][4]            = {w(x0), w(x1), w(x2) (0 if cin==2), bias}
//   per hidden layer l=1..L-2:
//     Wf [NT(mt)][NT(kt)][4(q)][64(lane)][4(j)]  A-fragments of W   : W[32mt+i][32kt+8q+4hi+j]
//     Wb [NT(mt)][NT(kt)][4(q)][64(lane)][4(j)]  A-fragments of W^T : W[32kt+8q+4hi+j][32mt+i]
//     bp [FP]
//   Whp [4][FP]  (rows >= cout zero),  bhp[4]
//   (lane = 32*hi + i ; this is the operand order of v_mfma_f32_32x32x2_f32, see brief_hip.hip)
// number of 32-feature tiles the width is padded to: exact up to 8 tiles, then 12 or 16 (only those
// kernel instantiations exist above 256 features)
BL_HD int brief_nt(const brief_siren_desc &d)
{
    const int nt = (d.features + 31) / 32;
    return nt <= 8 ? nt : (nt <= 12 ? 12 : 16);
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
BL_HD int64_t brief_pk_count(const brief_siren_desc &d)
{
    const int64_t FP = 32 * brief_nt(d);
    return brief_pk_head(d) + 4 * FP + 4;
}

// --- fused-kernel geometry per NT (how the 4 waves of a workgroup split features x sample tiles)
BL_HD int brief_wm(int nt) { return nt >= 3 ? 4 : nt; }          // waves along features
BL_HD int brief_ws(int nt) { return 4 / brief_wm(nt); }          // sample tiles (of 32) per workgroup
BL_HD int64_t brief_wg_samples(int nt) { return 32 * brief_ws(nt); }
BL_HD int64_t brief_npad(int nt, int64_t n)
{
    const int64_t g = brief_wg_samples(nt);
    return (n + g - 1) / g * g;
}
