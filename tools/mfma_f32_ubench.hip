// micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 under the dependency patterns the fused kernel uses
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int NACC, int RUN>   // NACC accumulators, RUN consecutive MFMAs on one accumulator before switching
__global__ void k(float *out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
#pragma unroll
                for (int r = 0; r < RUN; ++r) acc[i] = MFMA(a, b, acc[i]);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int RUN>
void run(const char *name, int threads, int blocks_per_cu)
{
    float *out;
    hipMalloc(&out, 256 * 8 * 1024 * sizeof(float));
    const int iters = 2000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, RUN>), dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0f, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, RUN>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)iters * 8 * NACC * RUN * (threads / 64) * blocks;
    const double tf = n_mfma * 4096 / (ms * 1e-3) / 1e12;
    // cycles per MFMA per SIMD assuming 2.4 GHz and waves spread over 4 SIMDs
    printf("%-28s threads=%4d blocks/CU=%d : %7.2f ms  %6.1f TFLOP/s\n", name, threads, blocks_per_cu, ms, tf);
    hipFree(out);
}

int main()
{
    run<1, 1>("1 acc, dependent chain", 256, 1);
    run<2, 1>("2 acc alternating", 256, 1);
    run<2, 4>("2 acc, runs of 4 (fused)", 256, 1);
    run<4, 1>("4 acc alternating", 256, 1);
    run<1, 1>("1 acc, dependent chain", 256, 2);
    run<2, 4>("2 acc, runs of 4 (fused)", 256, 2);
    run<2, 1>("2 acc alternating", 256, 2);
    run<4, 1>("4 acc alternating", 256, 2);
    run<2, 4>("2 acc, runs of 4", 512, 1);
    return 0;
}
