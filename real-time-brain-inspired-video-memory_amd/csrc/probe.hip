// vm_probe_mfma: what this GPU sustains on 16-bit MFMA work, measured on the box the bench runs on (include/vidmem.h,
// "measurement"; DESIGN.md 4.2 "A power-shaped ceiling").  The chip lowers its clock under matrix load
// (MI355X_MICROARCH.md, DVFS give-back), so the encoder GEMM's distance from the 2.5 PFLOP/s of the data sheet says
// little about the kernel; its distance from these loops does.  Three synthetic loops, each the K loop of
// gemm256p_kernel (gemm.hip) with more and more of its data movement added and nothing else - no epilogue, no barrier,
// no tile boundary, no cache miss, no store:
//   variant 0  registers only: 32 v_mfma_f32_16x16x32_f16 per iteration with the operand pattern of a 128 x 64 wave tile
//              (8 A x 4 B fragments, every MFMA sees other operands than the one before), two waves per SIMD, every CU
//   variant 1  + the wave tile's fragment reads: 12 conflict-free ds_read_b128 per 32 MFMAs from a rotating LDS image
//   variant 2  + the tile's staging: 4 x 1 KiB of LDS-DMA per wave and 32 MFMAs (64 KiB per workgroup and 64-deep K-tile
//              of a 256 x 256 tile) out of a 2 MiB buffer that stays in every L2, retired with a counted wait
// (the stand-alone developer version with more legs: tools/mfma_power_probe.hip)
#include "vm_common.h"

#include <vector>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int PROBE_SRC_VECS = 4096;              // 64 KiB of operand values
constexpr unsigned PROBE_POOL = 2u * 1024 * 1024; // the staging source of variant 2

template <int NR, int DMA>
__global__ __launch_bounds__(512, 1) void probe_loop(const h8 *src, float *sink, int iters, const char *pool) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (NR > 0) {
        for (int i = threadIdx.x; i < 96 * 1024 / 16; i += 512) reinterpret_cast<h8 *>(smem)[i] = src[i % PROBE_SRC_VECS];
        __syncthreads();
    }
    h8 f[12];
    for (int i = 0; i < 12; ++i) f[i] = src[(i * 64 + (threadIdx.x & 63)) % PROBE_SRC_VECS];
    f32x4 acc[8][4] = {};
    const int lane16 = (threadIdx.x & 63) * 16;
    for (int it = 0; it < iters; ++it) {
        const char *img = smem + (it & 7) * 12 * 1024 + lane16;
        if (DMA) {
            const int wave = threadIdx.x >> 6;
            const unsigned off = ((unsigned)(it * 4) * 8192u + wave * 1024u + blockIdx.x * 64u * 1024u) & (PROBE_POOL - 1);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(pool + ((off + u * 8192u) & (PROBE_POOL - 1)) + lane16),
                    (__attribute__((address_space(3))) void *)(smem + 96 * 1024 + wave * 4096 + u * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) f[i] = *reinterpret_cast<const h8 *>(img + i * 1024);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[i], f[8 + j], acc[i][j], 0, 0, 0);
        if (NR < 12) {   // what is not re-read rotates, so that every MFMA still sees changing operands
            const h8 t = f[NR];
#pragma unroll
            for (int i = NR; i < 11; ++i) f[i] = f[i + 1];
            f[11] = t;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123.456f) sink[0] = s;   // never true: keeps the accumulators live
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA in flight when the wave ends
}

}  // namespace

extern "C" int vm_probe_mfma(vm_ctx *ctx, int variant, int zero_operands, double seconds, double *tflops_host,
                             void *stream) {
    if (!ctx || !tflops_host || variant < 0 || variant > 2 || !(seconds > 0.0) || seconds > 30.0)
        return vm_fail(ctx, VM_ERR_INVALID, "vm_probe_mfma: variant 0..2, 0 < seconds <= 30");
    VM_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t src_bytes = (size_t)PROBE_SRC_VECS * 16, pool_bytes = PROBE_POOL + 4096;
    char *buf = nullptr;
    VM_HIP(ctx, hipMalloc((void **)&buf, src_bytes + pool_bytes + 256));
    int rc = VM_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto done = [&](int code) {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        (void)hipFree(buf);
        return code;
    };
    {
        std::vector<_Float16> h((src_bytes + pool_bytes) / 2);
        uint32_t lcg = 3u;   // fixed seed: uniform values in (-0.25, 0.25), the magnitude of normalised activations
        for (auto &v : h) {
            lcg = lcg * 1664525u + 1013904223u;
            v = zero_operands ? (_Float16)0.f : (_Float16)(((lcg >> 8) * (1.0f / 16777216.0f) * 2.f - 1.f) * 0.25f);
        }
        hipError_t he = hipMemcpy(buf, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        if (he == hipSuccess) he = hipMemset(buf + src_bytes + pool_bytes, 0, 256);
        if (he == hipSuccess) he = hipEventCreate(&e0);
        if (he == hipSuccess) he = hipEventCreate(&e1);
        if (he != hipSuccess) return done(vm_fail(ctx, VM_ERR_HIP, "vm_probe_mfma: setup: %s", hipGetErrorString(he)));
    }
    const h8 *src = reinterpret_cast<const h8 *>(buf);
    const char *pool = buf + src_bytes;
    float *sink = reinterpret_cast<float *>(buf + src_bytes + pool_bytes);
    const int iters = 20000;   // x 32 MFMAs x 16 KFLOP x 8 waves = 84 GFLOP per workgroup and launch (~13 ms)
    const int cus = ctx->num_cus;
    const double flop = (double)cus * 8 * iters * 32 * 2.0 * 16 * 16 * 32;
    const size_t lds = variant == 0 ? 0 : 128 * 1024;
    auto k0 = probe_loop<0, 0>;
    auto k1 = probe_loop<12, 0>;
    auto k2 = probe_loop<12, 1>;
    if (variant == 1) (void)hipFuncSetAttribute((const void *)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (variant == 2) (void)hipFuncSetAttribute((const void *)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    double spent_ms = 0.0;
    int launches = 0;
    // one launch per event pair and one wait per launch: the loop ends on the measured time, not on a guess
    while (spent_ms < seconds * 1e3 && launches < 100000) {
        hipError_t he = hipEventRecord(e0, st);
        if (he != hipSuccess) return done(vm_fail(ctx, VM_ERR_HIP, "vm_probe_mfma: %s", hipGetErrorString(he)));
        if (variant == 0) k0<<<cus, 512, 0, st>>>(src, sink, iters, pool);
        else if (variant == 1) k1<<<cus, 512, lds, st>>>(src, sink, iters, pool);
        else k2<<<cus, 512, lds, st>>>(src, sink, iters, pool);
        he = hipGetLastError();
        if (he == hipSuccess) he = hipEventRecord(e1, st);
        if (he == hipSuccess) he = hipEventSynchronize(e1);
        float ms = 0.f;
        if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
        if (he != hipSuccess) return done(vm_fail(ctx, VM_ERR_HIP, "vm_probe_mfma: %s", hipGetErrorString(he)));
        spent_ms += ms;
        ++launches;
    }
    *tflops_host = flop * launches / (spent_ms * 1e-3) / 1e12;
    return done(rc);
}
