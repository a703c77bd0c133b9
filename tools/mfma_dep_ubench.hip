// Does a chain of DEPENDENT f32 MFMAs (same accumulator back to back, as in k_fused's k-steps) issue at the full rate?
// NACC accumulators are rotated: NACC = 1 is the fully dependent chain, 2 is k_fused's (two M-tiles per wave) if the order alternated,
// 4 what the peak micro-benchmarks use.  One or two waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/mfma_dep_ubench.hip -o build/mfma_dep
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int RUN>
__global__ __launch_bounds__(512) void k(float *out, long long *ticks, int iters)
{
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float a[8], b[8];
    for (int u = 0; u < 8; ++u) { a[u] = 0.5f + 0.001f * (threadIdx.x + u); b[u] = 0.25f - 0.002f * (threadIdx.x % 17 + u); }
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            // RUN consecutive MFMAs on the same accumulator before moving to the next one
            const int t = (u / RUN) % NACC;
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[t], 0, 0, 0);
        }
    }
    const long long c1 = clock64();
    float r = 0.f;
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) r += acc[t][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) ticks[blockIdx.x] = c1 - c0;
}
template <int NACC, int RUN>
static void run(float *out, long long *t, int threads)
{
    const int iters = 20000;
    hipLaunchKernelGGL((k<NACC, RUN>), dim3(256), dim3(threads), 0, 0, out, t, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, RUN>), dim3(256), dim3(threads), 0, 0, out, t, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
    const double waves_per_simd = threads / 256.0;
    const double tf = 256.0 * (threads / 64) * iters * 8.0 * 4096 / (ms * 1e9);
    printf("  %d accumulator(s), runs of %d, %d wave(s) per SIMD: %.1f clock64 ticks per MFMA per wave; kernel %.3f ms = %.1f TFLOP/s\n", NACC, RUN, (int)waves_per_simd,
           s / 256 / (iters * 8.0), ms, tf);
}
int main()
{
    float *out; long long *t;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&t, 256 * 8);
    for (int threads : {256, 512}) {
        run<1, 1>(out, t, threads); run<2, 1>(out, t, threads); run<2, 4>(out, t, threads); run<4, 1>(out, t, threads);
    }
    return 0;
}
