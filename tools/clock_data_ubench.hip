// Does the clock a pure MFMA loop holds depend on the operand DATA?  clock_ubench.hip feeds constants (a = b = 1): no bit toggles
// between consecutive MFMAs.  Here the same loop runs on (0) constants, (1) per-lane random operands that change every MFMA,
// for the f32 32x32x2 and the bf16 32x32x16 instruction; the clock is shader cycles (clock64) per 100 MHz tick (wall_clock64).
//   hipcc --offload-arch=gfx950 -O3 tools/clock_data_ubench.hip -o build/clock_data && build/clock_data
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int RANDOM>
__global__ __launch_bounds__(256) void k32(float *out, long long *ticks, int iters)
{
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float a[8], b[8];
    for (int u = 0; u < 8; ++u) {
        const unsigned h1 = hash((blockIdx.x * 256 + threadIdx.x) * 16 + u), h2 = hash(h1 + 77);
        a[u] = RANDOM ? (int)(h1 >> 8) * (1.f / 8388608.f) - 1.f : 1.f;      // uniform in [-1, 1): every mantissa bit moves
        b[u] = RANDOM ? (int)(h2 >> 8) * (1.f / 8388608.f) - 1.f : 1.f;
    }
    const long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + j) & 7], acc[u & 3], 0, 0, 0);
    }
    const long long c1 = clock64(), r1 = wall_clock64();
    float r = 0.f;
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) r += acc[t][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = c1 - c0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int RANDOM>
__global__ __launch_bounds__(256) void k16(float *out, long long *ticks, int iters)
{
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    bf16x8 a[4], b[4];
    for (int u = 0; u < 4; ++u)
        for (int j = 0; j < 8; ++j) {
            const unsigned h1 = hash(((blockIdx.x * 256 + threadIdx.x) * 8 + u) * 8 + j), h2 = hash(h1 + 77);
            a[u][j] = (__bf16)(RANDOM ? (int)(h1 >> 8) * (1.f / 8388608.f) - 1.f : 1.f);
            b[u][j] = (__bf16)(RANDOM ? (int)(h2 >> 8) * (1.f / 8388608.f) - 1.f : 1.f);
        }
    const long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u & 3], b[(u + j) & 3], acc[u & 3], 0, 0, 0);
    }
    const long long c1 = clock64(), r1 = wall_clock64();
    float r = 0.f;
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) r += acc[t][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = c1 - c0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}

template <typename F>
static void run(const char *name, F kern, float *out, long long *t, int iters, double cyc_per_mfma, double flop_per_mfma)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 40; ++rep) {                // the last repetition is reported: ~0.8 s in, the power state has settled
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(1024), dim3(256), 0, 0, out, t, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    static long long h[2048]; (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    double lo = 1e9, hi = 0, sum = 0;
    for (int i = 0; i < 1024; ++i) { const double g = h[2 * i] / (h[2 * i + 1] * 10.0); lo = g < lo ? g : lo; hi = g > hi ? g : hi; sum += g; }
    const double mf = 1024.0 * 4 * iters * 8;
    printf("  %-34s %8.3f ms  clock mean %.3f GHz (min %.3f max %.3f)  %.1f TFLOP/s  MFMA cycles/MFMA %.2f\n", name, ms, sum / 1024, lo, hi,
           mf * flop_per_mfma / (ms * 1e9), h[0] / ((double)iters * 8));
    (void)cyc_per_mfma;
}
int main()
{
    float *out; long long *t;
    (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&t, 2048 * 8);
    const int iters = 30000;
    run("f32 32x32x2, constant operands", k32<0>, out, t, iters, 64, 32 * 32 * 2 * 2);
    run("f32 32x32x2, random operands", k32<1>, out, t, iters, 64, 32 * 32 * 2 * 2);
    run("bf16 32x32x16, constant operands", k16<0>, out, t, iters, 32, 32 * 32 * 16 * 2);
    run("bf16 32x32x16, random operands", k16<1>, out, t, iters, 32, 32 * 32 * 16 * 2);
    return 0;
}
