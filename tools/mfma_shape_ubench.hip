// f32-input MFMA shapes under sustained load: which one delivers more FLOP/s at the clock the chip holds?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_ubench.hip -o build/mfma_shape && build/mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE, int WPS>
__global__ __launch_bounds__(256 * WPS) void k(float *out, long long *ticks, int iters, const float *src)
{
    // operands from memory so that they are not compile-time constants; random-ish data (rule: bench on random operands)
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x * 8 + i) & 4095]; b[i] = src[(threadIdx.x * 8 + i + 2048) & 4095]; }
    float r = 0.f;
    const long long c0 = clock64(), r0 = wall_clock64();
    if (SHAPE == 32) {
        f32x16 acc[4];
        for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[u & 3], 0, 0, 0);
        }
        for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) r += acc[t][i];
    } else {
        f32x4 acc[8];
        for (int t = 0; t < 8; ++t) for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u & 7], b[u & 7], acc[u & 7], 0, 0, 0);
        }
        for (int t = 0; t < 8; ++t) for (int i = 0; i < 4; ++i) r += acc[t][i];
    }
    const long long c1 = clock64(), r1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = c1 - c0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int SHAPE, int WPS>
static void run(float *out, long long *t, const float *src)
{
    const int iters = 40000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, WPS>), dim3(256), dim3(256 * WPS), 0, 0, out, t, 100, src);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, WPS>), dim3(256), dim3(256 * WPS), 0, 0, out, t, iters, src);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    const double flop = (double)iters * 8 * 4096.0 * 256 * 4 * WPS;     // both bodies: 8 x 4096 FLOP (32x32x2) = 16 x 2048 FLOP per iteration
    printf("%s, %d wave(s)/SIMD: %.3f ms, %.1f TFLOP/s, clock %.3f GHz\n", SHAPE == 32 ? "32x32x2" : "16x16x4", WPS, ms, flop / ms / 1e9, (double)h[0] / h[1] * 0.1);
}
int main()
{
    float *out, *src; long long *t;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&t, 512 * 8); (void)hipMalloc(&src, 4096 * 4);
    float h[4096]; unsigned s = 12345u;
    for (int i = 0; i < 4096; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) / 8388608.0f) - 1.0f; }
    (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    run<32, 1>(out, t, src); run<16, 1>(out, t, src); run<32, 2>(out, t, src); run<16, 2>(out, t, src);
    return 0;
}
