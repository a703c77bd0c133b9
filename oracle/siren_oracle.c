/* oracle/siren_oracle.c — plain-C restatement of the reference algorithm for the hot path.
 * TEST INFRASTRUCTURE ONLY (see siren_oracle.h).  Citations are into /root/reference. */
#include "siren_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

size_t oracle_param_count(const oracle_desc *d)
{   /* utils/Networks.py:292-297 */
    const size_t F = (size_t)d->features;
    return (size_t)d->cin * F + F + (size_t)(d->layers - 2) * (F * F + F) + F * (size_t)d->cout + (size_t)d->cout;
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* torch-CPU linspace (ATen RangeFactories): step = (hi-lo)/(n-1) in f32;
 * i < n/2: lo + step*i ; else hi - step*(n-1-i), each a single fused rounding. */
void oracle_linspace(float lo, float hi, int64_t n, float *out)
{
    if (n == 1) { out[0] = lo; return; }
    const float step = (hi - lo) / (float)(n - 1);
    const int64_t half = n / 2;
    for (int64_t i = 0; i < n; ++i)
        out[i] = i < half ? fmaf(step, (float)i, lo) : fmaf(-step, (float)(n - 1 - i), hi);
}

void oracle_grid_coords(const int64_t *dims, int ndim, float lo, float hi, const int64_t *idx, int64_t n, float *out)
{   /* utils/dataset.py:36-60: stack(meshgrid(linspace...), -1) flattened '(d h w) c' */
    float *ax[3];
    for (int a = 0; a < ndim; ++a) {
        ax[a] = (float *)malloc(sizeof(float) * (size_t)dims[a]);
        oracle_linspace(lo, hi, dims[a], ax[a]);
    }
    for (int64_t s = 0; s < n; ++s) {
        int64_t v = idx ? idx[s] : s;
        for (int a = ndim - 1; a >= 0; --a) {
            out[s * ndim + a] = ax[a][v % dims[a]];
            v /= dims[a];
        }
    }
    for (int a = 0; a < ndim; ++a) free(ax[a]);
}

#define REAL float
#define SUF f32
#define SIN sinf
#define COS cosf
#include "siren_body.inc"
#undef REAL
#undef SUF
#undef SIN
#undef COS

#define REAL double
#define SUF f64
#define SIN sin
#define COS cos
#include "siren_body.inc"
#undef REAL
#undef SUF
#undef SIN
#undef COS

void oracle_optim_step(int kind, float *p, const float *g, float *s1, float *s2, int64_t n,
                       double lr, double b1, double b2, double eps, int64_t t)
{
    if (kind == 0) {
        /* torch/optim/adamax.py _single_tensor_adamax:
         *   exp_avg.lerp_(grad, 1-beta1); exp_inf = max(exp_inf*beta2, |grad|+eps);
         *   clr = lr/(1-beta1**t); param.addcdiv_(exp_avg, exp_inf, value=-clr) */
        const float w = (float)(1.0 - b1), fb2 = (float)b2, feps = (float)eps;
        const float nclr = (float)(-(lr / (1.0 - pow(b1, (double)t))));
        for (int64_t i = 0; i < n; ++i) {
            s1[i] = fmaf(w, g[i] - s1[i], s1[i]);
            const float a = s2[i] * fb2, b = fabsf(g[i]) + feps;
            s2[i] = a > b ? a : b;
            p[i] = p[i] + nclr * s1[i] / s2[i];
        }
    } else if (kind == 1) {
        /* torch/optim/adam.py _single_tensor_adam (amsgrad=False, weight_decay=0):
         *   exp_avg.lerp_(grad,1-beta1); exp_avg_sq.mul_(beta2).addcmul_(grad,grad,value=1-beta2)
         *   denom = (exp_avg_sq.sqrt()/sqrt(1-beta2**t)).add_(eps); param.addcdiv_(exp_avg, denom, value=-lr/(1-beta1**t)) */
        const float w = (float)(1.0 - b1), fb2 = (float)b2, w2 = (float)(1.0 - b2), feps = (float)eps;
        const float nss = (float)(-(lr / (1.0 - pow(b1, (double)t))));
        const float bc2s = (float)sqrt(1.0 - pow(b2, (double)t));
        for (int64_t i = 0; i < n; ++i) {
            s1[i] = fmaf(w, g[i] - s1[i], s1[i]);
            s2[i] = s2[i] * fb2 + w2 * (g[i] * g[i]);
            const float den = sqrtf(s2[i]) / bc2s + feps;
            p[i] = p[i] + nss * s1[i] / den;
        }
    } else {
        const float nlr = (float)(-lr);
        for (int64_t i = 0; i < n; ++i) p[i] = p[i] + nlr * g[i];
    }
}

void oracle_invnormalize(const float *yhat, int64_t n, float scale_min, float scale_max, double vmin, double vmax,
                         int dtype_code, void *out)
{   /* utils/io.py:136-147: data -= a; data /= (b-a); clip(0,1); data*(max-min)+min; np.array(.., dtype) */
    const float den = (float)((double)scale_max - (double)scale_min);
    const float span = (float)(vmax - vmin), fmin = (float)vmin;
    for (int64_t i = 0; i < n; ++i) {
        volatile float t = yhat[i] - scale_min;
        t = t / den;
        t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
        volatile float u = t * span;   /* separate roundings: torch does mul then add */
        u = u + fmin;
        if (dtype_code == 0) ((uint8_t *)out)[i] = (uint8_t)(int32_t)u;
        else ((uint16_t *)out)[i] = (uint16_t)(int32_t)u;
    }
}

void oracle_normalize(const void *src, int dtype_code, int64_t n, float scale_min, float scale_max, double vmin, double vmax, float *out)
{   /* utils/io.py:65-80: f32 array ops with python-float scalars (cast to f32 by numpy) */
    const float fmin = (float)vmin, den = (float)(vmax - vmin);
    const float sc = (float)((double)scale_max - (double)scale_min);
    for (int64_t i = 0; i < n; ++i) {
        const float v = dtype_code == 0 ? (float)((const uint8_t *)src)[i] : (float)((const uint16_t *)src)[i];
        volatile float t = v - fmin;
        t = t / den;
        t = t * sc;
        t = t + scale_min;
        out[i] = t;
    }
}
