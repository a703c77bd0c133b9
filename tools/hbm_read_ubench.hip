// achievable HBM streaming-read bandwidth on one MI355X: grid-stride 16-byte loads, U independent loads in flight per thread,
// G workgroups of 256 threads per CU.  hipcc --offload-arch=gfx950 -O3 tools/hbm_read_ubench.hip -o build/hbm_read_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int U>
__global__ __launch_bounds__(256) void k_read(const float4 *__restrict__ x, size_t n4, float *out)
{
    float4 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { acc[u].x += v[u].x; acc[u].y += v[u].y; acc[u].z += v[u].z; acc[u].w += v[u].w; }
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) s += acc[u].x + acc[u].y + acc[u].z + acc[u].w;
    if (s == 12345.678f) out[0] = s;
}
// k_wgrad_x3's pattern: workgroup b streams a private contiguous region; per iteration every thread has NL 16-byte loads in flight
// (NL * 8 KB per 512-thread workgroup), consumed an iteration later; optional workgroup barrier per iteration
template <int NL, bool BAR>
__global__ __launch_bounds__(512) void k_stream(const float4 *__restrict__ x, size_t region4, int iters, float *out)
{
    const float4 *p = x + (size_t)blockIdx.x * region4 + threadIdx.x;
    float4 cur[NL], acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < NL; ++u) cur[u] = p[u * 512];
    for (int it = 1; it <= iters; ++it) {
        const float4 *q = p + (size_t)(it < iters ? it : iters - 1) * (NL * 512);
        float4 nxt[NL];
#pragma unroll
        for (int u = 0; u < NL; ++u) nxt[u] = q[u * 512];
        if (BAR) __syncthreads();
#pragma unroll
        for (int u = 0; u < NL; ++u) { acc.x += cur[u].x; acc.y += cur[u].y; acc.z += cur[u].z; acc.w += cur[u].w; cur[u] = nxt[u]; }
    }
    const float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 12345.678f) out[0] = s;
}
template <int NL, bool BAR>
static void run_stream(const float4 *x, size_t n4, float *out, int wgs)
{
    const size_t region4 = n4 / wgs;
    const int iters = (int)(region4 / (NL * 512));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_stream<NL, BAR>), dim3(wgs), dim3(512), 0, 0, x, region4, iters, out);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_stream<NL, BAR>), dim3(wgs), dim3(512), 0, 0, x, region4, iters, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 20;
    printf("  private streams: %4d workgroups x 512 threads, %d x 8 KB in flight each, barrier %d, %d iterations: %.3f ms  %.2f TB/s\n", wgs, NL, (int)BAR, iters,
           ms, (double)wgs * iters * NL * 512 * 16.0 / ms / 1e9);
}
template <int U>
static void run(const float4 *x, size_t n4, float *out, int wgs)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_read<U>, dim3(wgs), dim3(256), 0, 0, x, n4, out);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_read<U>, dim3(wgs), dim3(256), 0, 0, x, n4, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 20;
    printf("  U=%d wgs=%5d (%2d per CU, %3d KB in flight per CU): %.3f ms  %.2f TB/s\n", U, wgs, wgs / 256, wgs / 256 * 256 * 16 * U / 1024, ms, n4 * 16.0 / ms / 1e9);
}
int main()
{
    for (double gb : {0.6, 2.0}) {
        const size_t n4 = (size_t)(gb * 1e9 / 16);
        float4 *x; float *out;
        hipMalloc(&x, n4 * 16); hipMalloc(&out, 4);
        hipMemset(x, 1, n4 * 16);
        printf("%.1f GB\n", gb);
        for (int per : {1, 2, 4, 8}) { run<1>(x, n4, out, 256 * per); run<2>(x, n4, out, 256 * per); run<4>(x, n4, out, 256 * per); run<8>(x, n4, out, 256 * per); }
        for (int wgs : {255, 510, 1020}) { run_stream<2, false>(x, n4, out, wgs); run_stream<4, false>(x, n4, out, wgs); run_stream<8, false>(x, n4, out, wgs); run_stream<8, true>(x, n4, out, wgs); }
        hipFree(x); hipFree(out);
    }
    return 0;
}
