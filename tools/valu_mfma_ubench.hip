// (1) do f32 MFMA and VALU FMA overlap on one SIMD?  (2) accuracy of v_sin_f32 / v_cos_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// waves [0,4) of a 512-thread block do MFMA, waves [4,8) do VALU fma chains (mode 0: both, 1: MFMA only, 2: VALU only)
__global__ void k_mix(float *out, int iters, int mode)
{
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (mode == 2) return;
        f32x16 acc0, acc1;
        for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
        float a = 1.0f + threadIdx.x * 1e-9f, b = 1.0f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc0 = MFMA(a, b, acc0); acc1 = MFMA(a, b, acc1); }
        }
        for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i];
    } else {
        if (mode == 1) return;
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
        const float c1 = 1.0000001f, c2 = 1e-7f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 64; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __fmaf_rn(v[i], c1, c2);    // 512 independent-ish FMAs per iteration
        }
        for (int i = 0; i < 8; ++i) r += v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ void k_sin(const float *f, float *s, float *c, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { s[i] = __builtin_amdgcn_sinf(f[i]); c[i] = __builtin_amdgcn_cosf(f[i]); }
}

static float timeit(int mode, int iters, float *out)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mix, dim3(256), dim3(512), 0, 0, out, 10, mode);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_mix, dim3(256), dim3(512), 0, 0, out, iters, mode);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float *out; (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    const int iters = 4000;
    const float t_both = timeit(0, iters, out), t_mfma = timeit(1, iters, out), t_valu = timeit(2, iters, out);
    printf("MFMA only %.3f ms (%.1f TF)   VALU only %.3f ms (%.1f TF)   both %.3f ms   sum %.3f  max %.3f\n", t_mfma,
           iters * 16.0 * 4096 * 4 * 256 / t_mfma / 1e9, t_valu, iters * 512.0 * 128 * 4 * 256 / t_valu / 1e9, t_both, t_mfma + t_valu,
           fmaxf(t_mfma, t_valu));
    // accuracy of the hardware sin/cos on revolutions in [-0.5, 0.5]
    const int n = 1 << 22;
    std::vector<float> hf(n), hs(n), hc(n);
    for (int i = 0; i < n; ++i) hf[i] = (float)((i + 0.5) / n - 0.5);
    float *df, *ds, *dc;
    (void)hipMalloc(&df, n * 4); (void)hipMalloc(&ds, n * 4); (void)hipMalloc(&dc, n * 4);
    (void)hipMemcpy(df, hf.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_sin, dim3(n / 256), dim3(256), 0, 0, df, ds, dc, n);
    (void)hipMemcpy(hs.data(), ds, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hc.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0;
    for (int i = 0; i < n; ++i) {
        const double x = 6.283185307179586476925 * (double)hf[i];
        es = fmax(es, fabs((double)hs[i] - sin(x)));
        ec = fmax(ec, fabs((double)hc[i] - cos(x)));
    }
    printf("v_sin_f32 max abs err %.3e   v_cos_f32 max abs err %.3e   (input in revolutions, [-0.5,0.5])\n", es, ec);
    return 0;
}
