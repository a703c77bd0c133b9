// brief_math.h — scalar math shared by the gfx950 kernels (and compiled on the host by
// tests/test_sincos_host.py to measure its accuracy against float64).
//
// The reference evaluates sin() through torch-CPU (SLEEF, ~1 ulp).  v_sin_f32 takes
// revolutions and, after the 1/(2*pi) pre-multiply, carries ~2e-6 rad of argument error at the
// |w0*z| ~ 30 rad this network produces — too coarse for the parity band,
// so the default path is a 3-term Cody-Waite reduction by pi/2 plus Cephes-class minimax
// polynomials on [-pi/4, pi/4]; abs error <= ~1.2e-7 for |x| < 1e4.
#pragma once

#if defined(__HIPCC__)
#define BRIEF_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define BRIEF_HD static inline
#endif

// sin(x) and cos(x) together (shared range reduction).
BRIEF_HD void brief_sincosf(float x, float *s, float *c)
{
    const float k = rintf(x * 0.63661977236758134308f);      // x * 2/pi
    // pi/2 = C1 + C2 + C3 with C1, C2 truncated to 13 significant bits: k*C1 and k*C2 are exact
    // for |k| < 2^11, so the only rounding of the reduced argument is its own representation.
    float r = fmaf(-k, 1.570556640625f, x);
    r = fmaf(-k, 0.000239670276641845703125f, r);
    r = fmaf(-k, 1.58932547122958567e-08f, r);
    const float r2 = r * r;
    // sin(r) ~ r + r^3 * P(r^2) ; cos(r) ~ 1 - r^2/2 + r^4 * Q(r^2)
    float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(r2, ps, -1.6666654611e-1f);
    const float sr = fmaf(r * r2, ps, r);
    float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(r2, pc, 4.166664568298827e-2f);
    const float cr = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
    const int q = (int)k & 3;
    const float ss = (q & 1) ? cr : sr;
    const float cc = (q & 1) ? sr : cr;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}

BRIEF_HD float brief_sinf(float x)
{
    float s, c;
    brief_sincosf(x, &s, &c);
    return s;
}

BRIEF_HD float brief_cosf(float x)
{
    float s, c;
    brief_sincosf(x, &s, &c);
    return c;
}

// ---------------------------------------------------------------------------------------------
// Hot-path variant.  Measured on MI355X (tools/valu_mfma_ubench.hip): v_mfma_f32_32x32x2_f32 and
// ordinary VALU instructions do NOT overlap on a SIMD (the f32 MFMA runs on the vector FMA lanes:
// MFMA-only 1.88 ms + VALU-only 2.60 ms = 4.61 ms together), so every VALU instruction in an
// epilogue is paid in matrix throughput.  v_sin_f32 / v_cos_f32 take REVOLUTIONS and are accurate to
// 1.25e-7 abs on [-0.5, 0.5]; what they lack is an accurate reduction.  This does the reduction in
// two fused steps with 1/(2 pi) split hi+lo (argument error <= 1.9e-7 rad for |x| <= 200) and then
// uses the hardware ops: 5 VALU + 1-2 transcendental instead of ~30 VALU.  Total |err| <= ~3.1e-7.
BRIEF_HD float brief_revolutions(float x)
{
    const float k = rintf(x * 0.15915494309189535f);
    float f = fmaf(x, 0.15915493667125702f, -k);
    return fmaf(x, 6.4206382432985265e-09f, f);
}
#if defined(__HIP_DEVICE_COMPILE__)
#define BRIEF_SIN_REV(f) __builtin_amdgcn_sinf(f)
#define BRIEF_COS_REV(f) __builtin_amdgcn_cosf(f)
#else
#define BRIEF_SIN_REV(f) ((float)sin(6.283185307179586476925 * (double)(f)))
#define BRIEF_COS_REV(f) ((float)cos(6.283185307179586476925 * (double)(f)))
#endif
BRIEF_HD float brief_fast_sinf(float x) { return BRIEF_SIN_REV(brief_revolutions(x)); }
BRIEF_HD float brief_fast_cosf(float x) { return BRIEF_COS_REV(brief_revolutions(x)); }
BRIEF_HD void brief_fast_sincosf(float x, float *s, float *c)
{
    const float f = brief_revolutions(x);
    *s = BRIEF_SIN_REV(f);
    *c = BRIEF_COS_REV(f);
}
