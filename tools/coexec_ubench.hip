// Issue model of gfx950: how do MFMA (f32-input 32x32x2 and bf16 32x32x16) and VALU / transcendental instructions share
// a SIMD?   hipcc --offload-arch=gfx950 -O3 tools/coexec_ubench.hip -o /tmp/coexec && /tmp/coexec
//  (1) same wave: one MFMA followed by K independent v_fma_f32 (or v_sin_f32), one wave per SIMD: cycles per MFMA vs K
//  (2) two waves per SIMD: waves 0-3 MFMA only, waves 4-7 VALU only: alone vs together
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>   // 0: f32 32x32x2, 1: bf16 32x32x16
__device__ __forceinline__ f32x16 mfma(f32x16 c, float a, float b, bf16x8 ah, bf16x8 bh)
{
    if constexpr (KIND == 0) return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
}

template <int KIND, int K, int TRANS>
__global__ __launch_bounds__(256) void k_same(float *out, int iters)
{
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float a = 1.0f + threadIdx.x * 1e-9f, b = 1.0f;
    bf16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(1.0f + i); bh[i] = (__bf16)1.0f; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    const float c1 = 1.0000001f, c2 = 1e-7f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u & 3] = mfma<KIND>(acc[u & 3], a, b, ah, bh);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (TRANS) v[(u * K + k) & 7] = __builtin_amdgcn_sinf(v[(u * K + k) & 7]);
                else v[(u * K + k) & 7] = __fmaf_rn(v[(u * K + k) & 7], c1, c2);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = 0.f;
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) r += acc[t][i];
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND, int TRANS>
__global__ __launch_bounds__(512) void k_cross(float *out, int iters, int mode)
{
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (mode == 2) return;
        f32x16 acc[4];
        for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        float a = 1.0f + threadIdx.x * 1e-9f, b = 1.0f;
        bf16x8 ah, bh;
        for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(1.0f + i); bh[i] = (__bf16)1.0f; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] = mfma<KIND>(acc[u & 3], a, b, ah, bh);
        }
        for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) r += acc[t][i];
    } else {
        if (mode == 1) return;
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
        const float c1 = 1.0000001f, c2 = 1e-7f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = TRANS ? __builtin_amdgcn_sinf(v[i]) : __fmaf_rn(v[i], c1, c2);    // 64 per iteration
        }
        for (int i = 0; i < 8; ++i) r += v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <typename F>
static float timeit(F launch)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    launch(20000);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int KIND, int K, int TRANS>
static void same(float *out, float base)
{
    const float ms = timeit([&](int it) { hipLaunchKernelGGL((k_same<KIND, K, TRANS>), dim3(256), dim3(256), 0, 0, out, it); });
    printf("  %s + %d %s per MFMA: %.3f ms  (x%.2f of MFMA only)\n", KIND ? "bf16 32x32x16" : "f32 32x32x2", K, TRANS ? "v_sin" : "v_fma", ms, base > 0 ? ms / base : 1.0f);
}

template <int KIND>
static void suite(float *out)
{
    const float base = timeit([&](int it) { hipLaunchKernelGGL((k_same<KIND, 0, 0>), dim3(256), dim3(256), 0, 0, out, it); });
    printf("%s: MFMA only %.3f ms for 160000 MFMAs per wave, one wave per SIMD -> %.1f ns per MFMA\n", KIND ? "bf16 32x32x16" : "f32 32x32x2", base, base * 1e6 / 160000);
    same<KIND, 1, 0>(out, base); same<KIND, 2, 0>(out, base); same<KIND, 4, 0>(out, base); same<KIND, 6, 0>(out, base);
    same<KIND, 8, 0>(out, base); same<KIND, 12, 0>(out, base); same<KIND, 16, 0>(out, base);
    same<KIND, 1, 1>(out, base); same<KIND, 2, 1>(out, base); same<KIND, 4, 1>(out, base);
    for (int tr = 0; tr < 2; ++tr) {
        float t[3];
        for (int mode = 0; mode < 3; ++mode) {
            if (tr) t[mode] = timeit([&](int it) { hipLaunchKernelGGL((k_cross<KIND, 1>), dim3(256), dim3(512), 0, 0, out, it, mode); });
            else t[mode] = timeit([&](int it) { hipLaunchKernelGGL((k_cross<KIND, 0>), dim3(256), dim3(512), 0, 0, out, it, mode); });
        }
        printf("  two waves per SIMD, %s: MFMA wave alone %.3f ms, %s wave alone %.3f ms (64 per 8 MFMAs), together %.3f ms (sum %.3f, max %.3f)\n",
               KIND ? "bf16" : "f32", t[1], tr ? "v_sin" : "v_fma", t[2], t[0], t[1] + t[2], t[1] > t[2] ? t[1] : t[2]);
    }
}

int main()
{
    float *out; (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    suite<0>(out);
    suite<1>(out);
    return 0;
}
