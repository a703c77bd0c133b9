// Does a VALU instruction issue while an MFMA executes on the same gfx950 SIMD?   Exact instruction streams (inline asm: the
// compiler cannot re-order them, which it did in coexec_ubench.hip's same-wave test), timed in shader cycles with s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/coissue_ubench.hip -o /tmp/coissue && /tmp/coissue
//  (1) one wave per SIMD:  { MFMA ; N x v_fma }   cycles per MFMA vs N         (f32 32x32x2 and bf16 32x32x16)
//  (2) one wave per SIMD:  { 8 MFMA ; 8N x v_fma } the burst shape of a GEMM chain followed by its epilogue
//  (3) two waves per SIMD: waves 0-3 MFMA only, waves 4-7 VALU only; priorities; each wave's own cycle count
//  (4) two waves per SIMD, both { 64 MFMA ; 512 VALU } out of phase (the k_fused shape: chain, then epilogue)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)
#define R32(x) R16(x) R16(x)
#define R64(x) R32(x) R32(x)

#define MF0 "v_mfma_f32_32x32x2_f32 a[0:15], %8, %9, a[0:15]\n"
#define MF1 "v_mfma_f32_32x32x2_f32 a[16:31], %8, %9, a[16:31]\n"
#define MF2 "v_mfma_f32_32x32x2_f32 a[32:47], %8, %9, a[32:47]\n"
#define MF3 "v_mfma_f32_32x32x2_f32 a[48:63], %8, %9, a[48:63]\n"
#define MB0 "v_mfma_f32_32x32x16_bf16 a[0:15], %10, %11, a[0:15]\n"
#define MB1 "v_mfma_f32_32x32x16_bf16 a[16:31], %10, %11, a[16:31]\n"
#define MB2 "v_mfma_f32_32x32x16_bf16 a[32:47], %10, %11, a[32:47]\n"
#define MB3 "v_mfma_f32_32x32x16_bf16 a[48:63], %10, %11, a[48:63]\n"
#define VF(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define V1 VF(0)
#define V2 VF(0) VF(1)
#define V4 VF(0) VF(1) VF(2) VF(3)
#define V8 V4 VF(4) VF(5) VF(6) VF(7)
#define V12 V8 V4
#define V16 V8 V8
#define V24 V16 V8
#define VS(i) "v_sin_f32 %" #i ", %" #i "\n"
#define S1 VS(0)
#define S2 VS(0) VS(1)
#define S4 VS(0) VS(1) VS(2) VS(3)
#define S8 S4 VS(4) VS(5) VS(6) VS(7)
#define NOFILL ""

#define CLOB "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15", \
    "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31", \
    "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47", \
    "a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define ASM_OPERANDS : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) \
                     : "v"(fa), "v"(fb), "v"(ah), "v"(bh) : CLOB

#define PROLOGUE                                                                                       \
    float v[8];                                                                                        \
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;                                        \
    float fa = 1.0f + threadIdx.x * 1e-3f, fb = 0.999f - threadIdx.x * 1e-4f;                          \
    bf16x8 ah, bh;                                                                                     \
    for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(1.0f + i + threadIdx.x * 0.01f); bh[i] = (__bf16)(0.5f - i * 0.1f); } \
    zero_acc();

__device__ __forceinline__ void zero_acc()
{
    asm volatile(
        "v_accvgpr_write_b32 a0, 0\n v_accvgpr_write_b32 a1, 0\n v_accvgpr_write_b32 a2, 0\n v_accvgpr_write_b32 a3, 0\n"
        "v_accvgpr_write_b32 a4, 0\n v_accvgpr_write_b32 a5, 0\n v_accvgpr_write_b32 a6, 0\n v_accvgpr_write_b32 a7, 0\n"
        "v_accvgpr_write_b32 a8, 0\n v_accvgpr_write_b32 a9, 0\n v_accvgpr_write_b32 a10, 0\n v_accvgpr_write_b32 a11, 0\n"
        "v_accvgpr_write_b32 a12, 0\n v_accvgpr_write_b32 a13, 0\n v_accvgpr_write_b32 a14, 0\n v_accvgpr_write_b32 a15, 0\n"
        "v_accvgpr_write_b32 a16, 0\n v_accvgpr_write_b32 a17, 0\n v_accvgpr_write_b32 a18, 0\n v_accvgpr_write_b32 a19, 0\n"
        "v_accvgpr_write_b32 a20, 0\n v_accvgpr_write_b32 a21, 0\n v_accvgpr_write_b32 a22, 0\n v_accvgpr_write_b32 a23, 0\n"
        "v_accvgpr_write_b32 a24, 0\n v_accvgpr_write_b32 a25, 0\n v_accvgpr_write_b32 a26, 0\n v_accvgpr_write_b32 a27, 0\n"
        "v_accvgpr_write_b32 a28, 0\n v_accvgpr_write_b32 a29, 0\n v_accvgpr_write_b32 a30, 0\n v_accvgpr_write_b32 a31, 0\n"
        "v_accvgpr_write_b32 a32, 0\n v_accvgpr_write_b32 a33, 0\n v_accvgpr_write_b32 a34, 0\n v_accvgpr_write_b32 a35, 0\n"
        "v_accvgpr_write_b32 a36, 0\n v_accvgpr_write_b32 a37, 0\n v_accvgpr_write_b32 a38, 0\n v_accvgpr_write_b32 a39, 0\n"
        "v_accvgpr_write_b32 a40, 0\n v_accvgpr_write_b32 a41, 0\n v_accvgpr_write_b32 a42, 0\n v_accvgpr_write_b32 a43, 0\n"
        "v_accvgpr_write_b32 a44, 0\n v_accvgpr_write_b32 a45, 0\n v_accvgpr_write_b32 a46, 0\n v_accvgpr_write_b32 a47, 0\n"
        "v_accvgpr_write_b32 a48, 0\n v_accvgpr_write_b32 a49, 0\n v_accvgpr_write_b32 a50, 0\n v_accvgpr_write_b32 a51, 0\n"
        "v_accvgpr_write_b32 a52, 0\n v_accvgpr_write_b32 a53, 0\n v_accvgpr_write_b32 a54, 0\n v_accvgpr_write_b32 a55, 0\n"
        "v_accvgpr_write_b32 a56, 0\n v_accvgpr_write_b32 a57, 0\n v_accvgpr_write_b32 a58, 0\n v_accvgpr_write_b32 a59, 0\n"
        "v_accvgpr_write_b32 a60, 0\n v_accvgpr_write_b32 a61, 0\n v_accvgpr_write_b32 a62, 0\n v_accvgpr_write_b32 a63, 0\n" ::: CLOB);
}

#define EPILOGUE(cyc, nm)                                                                              \
    float r = 0.f;                                                                                     \
    for (int i = 0; i < 8; ++i) r += v[i];                                                             \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                    \
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = (cyc);

// (1)/(2): 8 MFMAs per loop iteration, BODY is the full asm text of one iteration
#define DEF_SAME(name, BODY)                                                                           \
    __global__ __launch_bounds__(256) void name(float *out, long long *cycles, int iters)              \
    {                                                                                                  \
        PROLOGUE                                                                                       \
        const long long t0 = __builtin_amdgcn_s_memtime();                                             \
        for (int it = 0; it < iters; ++it) asm volatile(BODY ASM_OPERANDS);                            \
        const long long t1 = __builtin_amdgcn_s_memtime();                                             \
        EPILOGUE(t1 - t0, name)                                                                        \
    }

#define F8(FILL) MF0 FILL MF1 FILL MF2 FILL MF3 FILL MF0 FILL MF1 FILL MF2 FILL MF3 FILL
#define B8(FILL) MB0 FILL MB1 FILL MB2 FILL MB3 FILL MB0 FILL MB1 FILL MB2 FILL MB3 FILL
DEF_SAME(f_0, F8(NOFILL))
DEF_SAME(f_1, F8(V1))
DEF_SAME(f_2, F8(V2))
DEF_SAME(f_4, F8(V4))
DEF_SAME(f_8, F8(V8))
DEF_SAME(f_12, F8(V12))
DEF_SAME(f_16, F8(V16))
DEF_SAME(f_24, F8(V24))
DEF_SAME(f_s1, F8(S1))
DEF_SAME(f_s2, F8(S2))
DEF_SAME(f_s4, F8(S4))
DEF_SAME(f_s8, F8(S8))
DEF_SAME(b_0, B8(NOFILL))
DEF_SAME(b_1, B8(V1))
DEF_SAME(b_2, B8(V2))
DEF_SAME(b_4, B8(V4))
DEF_SAME(b_8, B8(V8))
DEF_SAME(b_12, B8(V12))
DEF_SAME(b_s1, B8(S1))
DEF_SAME(b_s2, B8(S2))
DEF_SAME(b_s4, B8(S4))
// bursts: 8 MFMAs, then all the filler
DEF_SAME(fb_4, F8(NOFILL) R8(V4))
DEF_SAME(fb_8, F8(NOFILL) R8(V8))
DEF_SAME(fb_16, F8(NOFILL) R8(V16))
// VALU alone (what the filler costs by itself)
DEF_SAME(v_8x8, R8(V8))
DEF_SAME(s_8x8, R8(S8))

// (3) two waves per SIMD: waves 0-3 run MFMAs, waves 4-7 VALU.  mode bit0: MFMA waves active, bit1: VALU waves active,
// bits 2-3: s_setprio of the VALU waves, bits 4-5: s_setprio of the MFMA waves, bit 6: transcendental filler
__global__ __launch_bounds__(512) void k_cross(float *out, long long *cycles, int iters, int mode)
{
    PROLOGUE
    const int wave = threadIdx.x >> 6;
    long long t0 = 0, t1 = 0;
    if (wave < 4) {
        if (mode & 1) {
            const int p = (mode >> 4) & 3;
            if (p == 1) __builtin_amdgcn_s_setprio(1); else if (p == 2) __builtin_amdgcn_s_setprio(2); else if (p == 3) __builtin_amdgcn_s_setprio(3);
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) asm volatile(F8(NOFILL) ASM_OPERANDS);
            t1 = __builtin_amdgcn_s_memtime();
        }
    } else {
        if (mode & 2) {
            const int p = (mode >> 2) & 3;
            if (p == 1) __builtin_amdgcn_s_setprio(1); else if (p == 2) __builtin_amdgcn_s_setprio(2); else if (p == 3) __builtin_amdgcn_s_setprio(3);
            t0 = __builtin_amdgcn_s_memtime();
            if (mode & 64) for (int it = 0; it < iters; ++it) asm volatile(R4(S8) ASM_OPERANDS);
            else for (int it = 0; it < iters; ++it) asm volatile(R8(V8) ASM_OPERANDS);
            t1 = __builtin_amdgcn_s_memtime();
        }
    }
    EPILOGUE(t1 - t0, k_cross)
}

// (3b) as (3) with bf16 MFMAs in waves 0-3.  mode bit 7: the VALU waves run a { 1 v_sin ; 1 v_fma } mix (a sine epilogue's shape)
__global__ __launch_bounds__(512) void k_cross16(float *out, long long *cycles, int iters, int mode)
{
    PROLOGUE
    const int wave = threadIdx.x >> 6;
    long long t0 = 0, t1 = 0;
    if (wave < 4) {
        if (mode & 1) {
            const int p = (mode >> 4) & 3;
            if (p == 1) __builtin_amdgcn_s_setprio(1); else if (p == 2) __builtin_amdgcn_s_setprio(2); else if (p == 3) __builtin_amdgcn_s_setprio(3);
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) asm volatile(B8(NOFILL) ASM_OPERANDS);
            t1 = __builtin_amdgcn_s_memtime();
        }
    } else {
        if (mode & 2) {
            const int p = (mode >> 2) & 3;
            if (p == 1) __builtin_amdgcn_s_setprio(1); else if (p == 2) __builtin_amdgcn_s_setprio(2); else if (p == 3) __builtin_amdgcn_s_setprio(3);
            t0 = __builtin_amdgcn_s_memtime();
            if (mode & 128) for (int it = 0; it < iters; ++it) asm volatile(R8(VS(0) VF(1) VS(2) VF(3)) ASM_OPERANDS);      // 16 v_sin + 16 v_fma
            else if (mode & 64) for (int it = 0; it < iters; ++it) asm volatile(R4(S8) ASM_OPERANDS);
            else for (int it = 0; it < iters; ++it) asm volatile(R8(V8) ASM_OPERANDS);
            t1 = __builtin_amdgcn_s_memtime();
        }
    }
    EPILOGUE(t1 - t0, k_cross16)
}

// (4b) two waves per SIMD, both { 64 bf16 MFMA ; 64 x (v_sin, v_fma) }: in phase / out of phase
__global__ __launch_bounds__(512) void k_phase16(float *out, long long *cycles, int iters, int mode)
{
    PROLOGUE
    const int wave = threadIdx.x >> 6;
    const bool second = wave >= 4;
    if (second && !(mode & 2)) return;
    if (!second && !(mode & 1)) return;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (second && (mode & 4)) asm volatile(R8(R8(VS(0) VF(1))) ASM_OPERANDS);
    for (int it = 0; it < iters; ++it) {
        asm volatile(R8(B8(NOFILL)) ASM_OPERANDS);
        asm volatile(R8(R8(VS(0) VF(1))) ASM_OPERANDS);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    EPILOGUE(t1 - t0, k_phase16)
}

// (4) two waves per SIMD, both { 64 MFMA ; 512 v_fma } ; waves 4-7 start with the VALU part (out of phase)
__global__ __launch_bounds__(512) void k_phase(float *out, long long *cycles, int iters, int mode)
{
    PROLOGUE
    const int wave = threadIdx.x >> 6;
    const bool second = wave >= 4;
    if (second && !(mode & 2)) return;
    if (!second && !(mode & 1)) return;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (second && (mode & 4)) asm volatile(R8(R8(V8)) ASM_OPERANDS);
    for (int it = 0; it < iters; ++it) {
        asm volatile(R8(F8(NOFILL)) ASM_OPERANDS);
        asm volatile(R8(R8(V8)) ASM_OPERANDS);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    EPILOGUE(t1 - t0, k_phase)
}

static float *d_out; static long long *d_cyc; static long long h_cyc[256 * 8];

template <typename F>
static void run(const char *name, F launch, int nwaves, double per, const char *unit)
{
    launch();
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemset(d_cyc, 0, sizeof(h_cyc));
    (void)hipEventRecord(e0);
    launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h_cyc, d_cyc, sizeof(h_cyc), hipMemcpyDeviceToHost);
    // block 7's waves (any block will do): cycles of waves 0 and (if present) 4
    const long long c0 = h_cyc[7 * nwaves + 0], c4 = nwaves > 4 ? h_cyc[7 * nwaves + 4] : 0;
    printf("%-34s %8.3f ms   wave0 %10lld cyc (%7.2f per %s)", name, ms, c0, c0 / per, unit);
    if (nwaves > 4) printf("   wave4 %10lld cyc (%7.2f)", c4, c4 / per);
    printf("\n");
}

int main()
{
    (void)hipMalloc(&d_out, 256 * 512 * sizeof(float));
    (void)hipMalloc(&d_cyc, sizeof(h_cyc));
    const int it = 20000;
    // warm the clocks
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(f_0, dim3(256), dim3(256), 0, 0, d_out, d_cyc, it);
    (void)hipDeviceSynchronize();
#define SAME(k) run(#k, [&] { hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d_out, d_cyc, it); }, 4, it * 8.0, "MFMA");
    printf("(1) one wave per SIMD, { MFMA ; N fillers } x 8 per iteration; f_N = f32 32x32x2 + N v_fma, f_sN = + N v_sin, b_ = bf16 32x32x16\n");
    SAME(f_0) SAME(f_1) SAME(f_2) SAME(f_4) SAME(f_8) SAME(f_12) SAME(f_16) SAME(f_24) SAME(f_s1) SAME(f_s2) SAME(f_s4) SAME(f_s8)
    SAME(b_0) SAME(b_1) SAME(b_2) SAME(b_4) SAME(b_8) SAME(b_12) SAME(b_s1) SAME(b_s2) SAME(b_s4)
    printf("(2) bursts: 8 f32 MFMAs then 8 x N v_fma\n");
    SAME(fb_4) SAME(fb_8) SAME(fb_16)
    printf("    filler alone: 64 v_fma / 64 v_sin per iteration (cycles per group of 8)\n");
    SAME(v_8x8) SAME(s_8x8)
    printf("(3) two waves per SIMD: waves 0-3 f32 MFMA (8 per iteration), waves 4-7 VALU (64 v_fma or 32 v_sin per iteration)\n");
    struct { const char *n; int mode; } cr[] = {
        {"mfma alone", 1}, {"v_fma alone", 2}, {"both", 3}, {"both, valu prio 3", 3 | (3 << 2)}, {"both, mfma prio 3", 3 | (3 << 4)},
        {"v_sin alone", 2 | 64}, {"both sin", 3 | 64}, {"both sin, valu prio 3", 3 | 64 | (3 << 2)}, {"both sin, mfma prio 3", 3 | 64 | (3 << 4)}};
    for (auto &c : cr) run(c.n, [&] { hipLaunchKernelGGL(k_cross, dim3(256), dim3(512), 0, 0, d_out, d_cyc, it, c.mode); }, 8, it * 8.0, "8");
    printf("(4) two waves per SIMD, both { 64 MFMA ; 512 v_fma }\n");
    struct { const char *n; int mode; } ph[] = {{"first half alone", 1}, {"both in phase", 3}, {"both out of phase", 7}};
    for (auto &c : ph) run(c.n, [&] { hipLaunchKernelGGL(k_phase, dim3(256), dim3(512), 0, 0, d_out, d_cyc, it / 8, c.mode); }, 8, it / 8 * 64.0, "MFMA");
    printf("(3b) two waves per SIMD: waves 0-3 bf16 MFMA (8 per iteration), waves 4-7 VALU (64 v_fma | 32 v_sin | 16 v_sin + 16 v_fma per iteration)\n");
    struct { const char *n; int mode; } cr16[] = {
        {"bf16 mfma alone", 1}, {"v_fma alone", 2}, {"both", 3}, {"both, valu prio 3", 3 | (3 << 2)}, {"both, mfma prio 3", 3 | (3 << 4)},
        {"v_sin alone", 2 | 64}, {"both sin", 3 | 64}, {"both sin, valu prio 3", 3 | 64 | (3 << 2)},
        {"sin+fma mix alone", 2 | 128}, {"both mix", 3 | 128}, {"both mix, valu prio 3", 3 | 128 | (3 << 2)}, {"both mix, mfma prio 3", 3 | 128 | (3 << 4)}};
    for (auto &c : cr16) run(c.n, [&] { hipLaunchKernelGGL(k_cross16, dim3(256), dim3(512), 0, 0, d_out, d_cyc, it, c.mode); }, 8, it * 8.0, "8");
    printf("(4b) two waves per SIMD, both { 64 bf16 MFMA ; 64 x (v_sin, v_fma) }\n");
    struct { const char *n; int mode; } ph16[] = {{"first half alone", 1}, {"both in phase", 3}, {"both out of phase", 7}};
    for (auto &c : ph16) run(c.n, [&] { hipLaunchKernelGGL(k_phase16, dim3(256), dim3(512), 0, 0, d_out, d_cyc, it / 8, c.mode); }, 8, it / 8 * 64.0, "MFMA");
    return 0;
}
