// Does the MFMA SHAPE change what the chip sustains under its power ceiling?  Register-only loops (no LDS, no memory in
// the timed part) over the two dense 16-bit shapes of gfx950, same FLOPs, same operand toggling pattern as a
// 128 x 64 wave tile of the persistent GEMM (8 A fragments x 4 B fragments per k-step, every MFMA sees other operands
// than the one before it), two waves per SIMD on every CU.
//   v_mfma_f32_16x16x32_f16: 16 KFLOP, A 4 + B 4 operand registers, 4 accumulators read and written
//   v_mfma_f32_32x32x16_f16: 32 KFLOP, A 4 + B 4 operand registers, 16 accumulators read and written
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_power_probe.hip -o tools/bin/mfma_power_probe
// run:   tools/bin/mfma_power_probe [seconds per leg]     (ZERO=1: all-zero operands; LDS=1: legs with fragment reads)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 vec8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(512, 1) void loop16(const vec8 *src, float *sink, int iters) {
    vec8 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = src[(i * 64 + (threadIdx.x & 63)) % 4096];
    for (int i = 0; i < 4; ++i) b[i] = src[((8 + i) * 64 + (threadIdx.x & 63)) % 4096];
    f32x4 acc[8][4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        // keep the operands live and changing a little: rotate the fragments (register moves only every 32 MFMAs)
        vec8 t = a[0];
#pragma unroll
        for (int i = 0; i < 7; ++i) a[i] = a[i + 1];
        a[7] = t;
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123.456f) sink[0] = s;
}

__global__ __launch_bounds__(512, 1) void loop32(const vec8 *src, float *sink, int iters) {
    // the same 128 x 64 x 32 per iteration: 4 x 2 tiles of 32 x 32, two k-steps of 16 -> 16 MFMAs of 32 KFLOP
    vec8 a[2][4], b[2][2];
    for (int k = 0; k < 2; ++k) {
        for (int i = 0; i < 4; ++i) a[k][i] = src[((k * 4 + i) * 64 + (threadIdx.x & 63)) % 4096];
        for (int i = 0; i < 2; ++i) b[k][i] = src[((8 + k * 2 + i) * 64 + (threadIdx.x & 63)) % 4096];
    }
    f32x16 acc[4][2] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k][i], b[k][j], acc[i][j], 0, 0, 0);
        vec8 t = a[0][0];
#pragma unroll
        for (int i = 0; i < 3; ++i) a[0][i] = a[0][i + 1];
        a[0][3] = a[1][0];
#pragma unroll
        for (int i = 0; i < 3; ++i) a[1][i] = a[1][i + 1];
        a[1][3] = t;
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    if (s == 123.456f) sink[0] = s;
}

// loop16 plus the fragment reads of the GEMM: NR of the 12 fragments of a k-step (8 A + 4 B) come from LDS as
// ds_read_b128 (conflict-free rows of 1 KiB, a different 12 KiB image every iteration), the others stay in registers.
// NR = 12: the 128 x 64 wave tile of gemm256p (24 reads per 64 MFMAs); NR = 8: what a 128 x 128 wave tile would read
// per FLOP (16 reads per 64 MFMAs); NR = 0: loop16.
// DMA = 1 adds the GEMM's staging: 4 LDS-DMA instructions of 1 KiB per wave and 32 MFMAs (64 KiB per workgroup and
// 64-deep K-tile) out of a 2 MiB buffer that stays in every L2, retired with a counted wait.
template <int NR, int DMA>
__global__ __launch_bounds__(512, 1) void loop16_lds(const vec8 *src, float *sink, int iters, const char *pool) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 96 * 1024 / 16; i += 512) reinterpret_cast<vec8 *>(smem)[i] = src[i % 4096];
    __syncthreads();
    vec8 f[12];
    for (int i = 0; i < 12; ++i) f[i] = src[(i * 64 + (threadIdx.x & 63)) % 4096];
    f32x4 acc[8][4] = {};
    const int lane16 = (threadIdx.x & 63) * 16;
    for (int it = 0; it < iters; ++it) {
        const char *img = smem + (it & 7) * 12 * 1024 + lane16;
        if (DMA) {
            const int wave = threadIdx.x >> 6;
            const unsigned off = ((unsigned)(it * 4) * 8192u + wave * 1024u + blockIdx.x * 64u * 1024u) & (2u * 1024 * 1024 - 1);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pool + ((off + u * 8192u) & (2u * 1024 * 1024 - 1)) + lane16),
                                                 (__attribute__((address_space(3))) void *)(smem + 96 * 1024 + wave * 4096 + u * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) f[i] = *reinterpret_cast<const vec8 *>(img + i * 1024);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[i], f[8 + j], acc[i][j], 0, 0, 0);
        if (NR < 12) {   // rotate what is not reloaded so that every MFMA still sees changing operands
            vec8 t = f[NR];
#pragma unroll
            for (int i = NR; i < 11; ++i) f[i] = f[i + 1];
            f[11] = t;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123.456f) sink[0] = s;
}

int main(int argc, char **argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    const bool zero = getenv("ZERO") != nullptr;
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    std::vector<_Float16> h(4096 * 8);
    srand(3);
    for (auto &v : h) v = zero ? (_Float16)0.f : (_Float16)((rand() / (float)RAND_MAX * 2.f - 1.f) * 0.25f);
    vec8 *src; float *sink;
    CK(hipMalloc(&src, h.size() * 2)); CK(hipMalloc(&sink, 4));
    CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    char *pool; CK(hipMalloc(&pool, 2u * 1024 * 1024 + 4096));
    { std::vector<_Float16> hp(1024 * 1024 + 2048); for (auto &v : hp) v = zero ? (_Float16)0.f : (_Float16)((rand() / (float)RAND_MAX * 2.f - 1.f) * 0.25f);
      CK(hipMemcpy(pool, hp.data(), hp.size() * 2, hipMemcpyHostToDevice)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;                                   // 20000 x 32 x 16 KFLOP x 8 waves = 84 GFLOP per workgroup
    const double flop = (double)cus * 8 * iters * 32 * 2.0 * 16 * 16 * 32;
    CK(hipFuncSetAttribute((const void *)loop16_lds<12, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipFuncSetAttribute((const void *)loop16_lds<8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipFuncSetAttribute((const void *)loop16_lds<12, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    const bool lds = getenv("LDS") != nullptr;   // LDS=1: the fragment-read legs instead of the shape legs
    const char *names[5] = {"16x16x32", "32x32x16", "16x16x32 + 12 ds_read_b128 / 32 MFMA", "16x16x32 + 8 ds_read_b128 / 32 MFMA",
                            "16x16x32 + 12 ds_read_b128 + 4 KiB LDS-DMA from L2 / 32 MFMA"};
    for (int round = 0; round < 3; ++round)
        for (int which = lds ? 2 : 0; which < (lds ? 5 : 2); ++which) {
            double spent = 0, best = 1e9, sum = 0; int n = 0;
            while (spent < secs) {
                CK(hipEventRecord(e0, 0));
                if (which == 0) loop16<<<cus, 512>>>(src, sink, iters);
                else if (which == 1) loop32<<<cus, 512>>>(src, sink, iters);
                else if (which == 2) loop16_lds<12, 0><<<cus, 512, 128 * 1024>>>(src, sink, iters, pool);
                else if (which == 3) loop16_lds<8, 0><<<cus, 512, 128 * 1024>>>(src, sink, iters, pool);
                else loop16_lds<12, 1><<<cus, 512, 128 * 1024>>>(src, sink, iters, pool);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                spent += ms * 1e-3; sum += ms; ++n; if (ms < best) best = ms;
            }
            printf("round %d %s%s: %d launches, mean %.2f ms = %.0f TFLOP/s (best %.2f ms = %.0f)\n", round,
                   names[which], zero ? " ZERO" : "", n, sum / n, flop / (sum / n * 1e-3) / 1e12, best,
                   flop / (best * 1e-3) / 1e12);
            fflush(stdout);
        }
    return 0;
}
