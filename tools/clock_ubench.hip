// What do clock64() (s_memtime) and wall_clock64() (s_memrealtime, 100 MHz) say around a pure f32-MFMA loop whose length in
// shader cycles is known (N x 64)?   hipcc --offload-arch=gfx950 -O3 tools/clock_ubench.hip -o build/clock && build/clock
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float *out, long long *ticks, int iters)
{
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float a = 1.0f + threadIdx.x * 1e-9f, b = 1.0f;
    const long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
    }
    float r = 0.f;
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) r += acc[t][i];
    const long long c1 = clock64(), r1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = c1 - c0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}
int main()
{
    float *out; long long *t;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&t, 512 * 8);
    const int iters = 40000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, t, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, t, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[512]; (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    const double mfma_cycles = (double)iters * 8 * 64;
    printf("event time %.3f ms; block 0: clock64 ticks %lld, wall_clock64 ticks %lld, MFMA cycles %.0f\n", ms, h[0], h[1], mfma_cycles);
    printf("  wall_clock64 rate %.1f MHz (vs event time); clock64 rate %.3f GHz (vs event time); clock64 ticks per MFMA cycle %.4f; MFMA cycles / event time = %.3f GHz\n",
           h[1] / (ms * 1e3), h[0] / (ms * 1e6), h[0] / mfma_cycles, mfma_cycles / (ms * 1e6));
    return 0;
}
