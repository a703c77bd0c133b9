// Launch-bound loops: N dependent tiny kernels on one stream, launched one by one against the same chain captured once in a
// hipGraph and replayed.  Reports host time to enqueue and GPU time to drain, per kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/graph_ubench.hip -o build/graph_ubench && build/graph_ubench
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
__global__ void k_tiny(float *p, int work)
{
    float v = p[threadIdx.x];
    for (int i = 0; i < work; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    float *p; (void)hipMalloc(&p, 4096); (void)hipMemset(p, 0, 4096);
    hipStream_t s; (void)hipStreamCreate(&s);
    const int K = 100, reps = 50;
    for (int work : {1, 2000}) {
        for (int blocks : {1, 512}) {
            for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(256), 0, s, p, work);
            (void)hipStreamSynchronize(s);
            double t0 = now();
            for (int r = 0; r < reps; ++r) for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(256), 0, s, p, work);
            double t1 = now();
            (void)hipStreamSynchronize(s);
            double t2 = now();
            hipGraph_t g; hipGraphExec_t ge;
            (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
            for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(256), 0, s, p, work);
            (void)hipStreamEndCapture(s, &g);
            (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
            double t3 = now();
            for (int r = 0; r < reps; ++r) (void)hipGraphLaunch(ge, s);
            double t4 = now();
            (void)hipStreamSynchronize(s);
            double t5 = now();
            printf("work %4d blocks %3d: stream launches: host %.2f us/kernel, total %.2f us/kernel | graph of %d replayed: host %.2f us/kernel, total %.2f us/kernel\n",
                   work, blocks, (t1 - t0) * 1e6 / (K * reps), (t2 - t0) * 1e6 / (K * reps), K, (t4 - t3) * 1e6 / (K * reps), (t5 - t3) * 1e6 / (K * reps));
            (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        }
    }
    return 0;
}
