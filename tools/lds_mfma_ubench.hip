// Do LDS fragment reads (ds_read_b128) and MFMAs overlap on a CU?  One workgroup of 8 waves per CU: waves 0-3 (one per SIMD) run
// an MFMA loop, waves 4-7 (the second wave of each SIMD) run a ds_read_b128 loop; each alone, then together.  Also the same-wave
// interleave (one wave per SIMD doing R reads per M MFMAs).   hipcc --offload-arch=gfx950 -O3 tools/lds_mfma_ubench.hip -o build/lds_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int BF16>
__global__ __launch_bounds__(512) void k(float *out, int mfma_iters, int lds_iters, int mode)
{
    __shared__ float4 lds[4096];                                  // 64 KB
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = make_float4(i, 1.f, 2.f, 3.f);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wave < 4 && (mode & 1)) {
        float a = 1.0f + lane * 1e-3f, b = 0.5f;
        bf16x8 a8, b8;
        for (int j = 0; j < 8; ++j) { a8[j] = (__bf16)(0.01f * (lane + j)); b8[j] = (__bf16)(0.5f - 0.01f * j); }
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (BF16) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[u & 3], 0, 0, 0);
                else acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
            }
        }
    }
    if (wave >= 4 && (mode & 2)) {
        // eight reads per iteration at immediate offsets from one address register, no VALU instruction in the loop
        const unsigned addr = (unsigned)(size_t)lds + (lane + (wave - 4) * 64) * 16;
        float4 v0, v1, v2, v3, v4, v5, v6, v7;
        for (int it = 0; it < lds_iters; ++it) {
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:4096\n\tds_read_b128 %2, %8 offset:8192\n\tds_read_b128 %3, %8 offset:12288\n\t"
                         "ds_read_b128 %4, %8 offset:16384\n\tds_read_b128 %5, %8 offset:20480\n\tds_read_b128 %6, %8 offset:24576\n\tds_read_b128 %7, %8 offset:28672\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(addr) : "memory");
        }
        s.x = v0.x + v1.x + v2.x + v3.x + v4.y + v5.y + v6.z + v7.w;
    }
    float r = s.x + s.y + s.z + s.w;
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) r += acc[t][i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int BF16>
static float run(float *out, int mi, int li, int mode)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<BF16>), dim3(256), dim3(512), 0, 0, out, mi, li, mode);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}
int main()
{
    float *out; (void)hipMalloc(&out, 256 * 512 * 4);
    const int mi = 20000;
    for (int bf = 0; bf < 2; ++bf) {
        // size the LDS loop to take about as long as the MFMA loop
        const float tm = bf ? run<1>(out, mi, 0, 1) : run<0>(out, mi, 0, 1);
        int li = 20000;
        float tl = bf ? run<1>(out, 0, li, 2) : run<0>(out, 0, li, 2);
        li = (int)(li * tm / tl);
        tl = bf ? run<1>(out, 0, li, 2) : run<0>(out, 0, li, 2);
        const float tb = bf ? run<1>(out, mi, li, 3) : run<0>(out, mi, li, 3);
        const double bytes = 256.0 * 4 * 64 * 16.0 * 8 * li;
        printf("%s MFMA waves alone %.3f ms; ds_read_b128 waves alone %.3f ms (%.1f B/clk/CU at 2.4 GHz); together %.3f ms (sum %.3f, max %.3f)\n",
               bf ? "bf16 32x32x16:" : "f32 32x32x2:  ", tm, tl, bytes / 256 / (tl * 1e-3 * 2.4e9), tb, tm + tl, tm > tl ? tm : tl);
    }
    return 0;
}
