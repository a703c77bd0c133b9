// What pulls the clock down under a matrix-bound loop?  f32 MFMA (32x32x2) with, per group of 4 MFMAs and wave:
//   L2: one 16-byte-per-lane load from a 1.5 MB L2-resident buffer (k_fused's weight-fragment stream), LDS: one ds_read_b128,
//   ST: one dword-per-lane store to a large HBM buffer every 8 MFMAs (k_fused's stash stores), VALU: 8 v_fma + 1 v_sin.
//   hipcc --offload-arch=gfx950 -O3 tools/power_ubench.hip -o build/power && build/power
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int L2, int LDS, int ST, int VALU>
__global__ __launch_bounds__(512) void k(float *out, long long *ticks, int iters, const float4 *w, float *stash, int nw4)
{
    __shared__ float4 sh[2048];
    for (int i = threadIdx.x; i < 2048; i += 512) sh[i] = w[i];
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float4 a = w[threadIdx.x], b = w[threadIdx.x + 512];
    float v = threadIdx.x * 1e-3f;
    int wo = (blockIdx.x * 977 + threadIdx.x) % nw4;
    int so = threadIdx.x;
    const long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (L2) { a = w[wo]; wo += 512; if (wo >= nw4) wo -= nw4; }
        if (LDS) b = sh[(it * 64 + threadIdx.x) & 2047];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[3], 0, 0, 0);
        if (ST && (it & 1)) { stash[(size_t)blockIdx.x * (1 << 20) + (so & ((1 << 20) - 1))] = acc[it & 3][it & 15]; so += 512; }
        if (VALU) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v = __fmaf_rn(v, 1.0000001f, 1e-7f);
            v = __builtin_amdgcn_sinf(v);
        }
    }
    float r = v;
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) r += acc[t][i];
    const long long c1 = clock64(), r1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = c1 - c0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int L2, int LDS, int ST, int VALU>
static void run(float *out, long long *t, const float4 *w, float *stash, int nw4)
{
    const int iters = 80000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<L2, LDS, ST, VALU>), dim3(256), dim3(512), 0, 0, out, t, 100, w, stash, nw4);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<L2, LDS, ST, VALU>), dim3(256), dim3(512), 0, 0, out, t, iters, w, stash, nw4);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    const double flop = (double)iters * 4 * 4096.0 * 256 * 8;
    printf("L2 %d  LDS %d  ST %d  VALU %d : %.3f ms, %.1f TFLOP/s, clock %.3f GHz, MFMA share of cycles %.3f\n", L2, LDS, ST, VALU, ms, flop / ms / 1e9,
           (double)h[0] / h[1] * 0.1, (double)iters * 4 * 64 * 2 / (double)h[0]);
}
int main()
{
    float *out, *stash; long long *t; float4 *w;
    const int nw4 = 98304;      // 1.5 MB of float4
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&t, 512 * 8); (void)hipMalloc(&w, nw4 * 16); (void)hipMalloc(&stash, (size_t)256 * (1 << 20) * 4);
    float *h = new float[nw4 * 4]; unsigned s = 12345u;
    for (int i = 0; i < nw4 * 4; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) / 8388608.0f) - 1.0f; }
    (void)hipMemcpy(w, h, (size_t)nw4 * 16, hipMemcpyHostToDevice);
    run<0, 0, 0, 0>(out, t, w, stash, nw4);
    run<1, 0, 0, 0>(out, t, w, stash, nw4);
    run<0, 1, 0, 0>(out, t, w, stash, nw4);
    run<1, 1, 0, 0>(out, t, w, stash, nw4);
    run<1, 1, 1, 0>(out, t, w, stash, nw4);
    run<1, 1, 0, 1>(out, t, w, stash, nw4);
    run<1, 1, 1, 1>(out, t, w, stash, nw4);
    return 0;
}
