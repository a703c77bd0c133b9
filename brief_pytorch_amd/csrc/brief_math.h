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
