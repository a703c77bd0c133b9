// resident workgroups per CU the runtime reports for a 256-thread kernel with `lds` bytes of dynamic LDS: hipcc --offload-arch=gfx950 tools/occupancy_probe.hip -o /tmp/occ && /tmp/occ
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256, 2) void k(float *o) { extern __shared__ float s[]; s[threadIdx.x] = 1.f; __syncthreads(); o[threadIdx.x] = s[255 - threadIdx.x]; }
int main()
{
    const int sizes[] = {65536, 81920, 81936, 83968, 98304, 131072, 163840};
    for (int lds : sizes) {
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        int nb = -1;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, lds);
        printf("lds %6d bytes: %d workgroups per CU (%s)\n", lds, nb, hipGetErrorString(e));
    }
    return 0;
}
