// Dense layers of the vision encoder: out[t, f] = sum_k X[t, k] * W[f, k] (+ fused epilogue), 16-bit operands,
// fp32 MFMA accumulation.  MFMA-bound: 95.9 % (ViT-B/16) / 91.4 % (CLIP-L/14-336) of the encoder's FLOPs run here.
//
// Tile 128 (features) x 128 (tokens) x 64 (k) per 256-thread workgroup, 4 waves as 2 x 2, each wave 64 x 64 =
// 4 x 4 tiles of v_mfma_f32_16x16x32.  The WEIGHT tile is the MFMA A operand and the ACTIVATION tile the B
// operand, so a lane ends up with 4 consecutive features of one token: epilogue stores are 8-byte (16-bit out)
// or 16-byte (fp32 residual) row pieces with no LDS transpose.
// Staging: global_load_lds_dwordx4 (16 B/lane, 1 KiB per wave instruction = 8 rows x 128 B) into a linear LDS
// image, double buffered; the XOR swizzle chunk ^= (row & 7) is applied on the per-lane SOURCE address and again
// on the ds_read_b128 address (both sides or neither), which makes the fragment reads bank-conflict free.
// Workgroups are renumbered so that the column tiles of one token panel land on the same XCD (its L2 then serves
// the panel's re-reads).
#include "vm_internal.h"
#include "vm_kernels.h"

namespace {

constexpr int BM = 128;  // tokens per tile
constexpr int BN = 128;  // features per tile
constexpr int BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;  // one operand tile, 16 KiB

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }

template <int DT, int EPI>
__global__ void __launch_bounds__(256, 2) gemm_kernel(GemmArgs g) {
    using E = vm_elem<DT>;
    using vec8 = typename E::vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][W tile | X tile]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, h = lane >> 4;
    const int tiles_n = g.N / BN;
    // XCD-aware renumbering (bijective for any grid size): ids that share (blockIdx % 8) become neighbours
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, qd = nwg >> 3, rm = nwg & 7;
    const int bid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (orig >> 3);
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int t0 = tm * BM, f0 = tn * BN;
    const int wf = wave & 1, wt = wave >> 1;

    const uint16_t *W = g.W, *X = g.X;
    const int K = g.K, M = g.M;

    // per-lane staging sources: wave w stages rows [32w, 32w+32) of both tiles, 4 instructions x 8 rows
    const int srow = lane >> 3, scp = lane & 7;
    const uint16_t *wsrc[4], *xsrc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int row = wave * 32 + u * 8 + srow;
        const int chunk = scp ^ (row & 7);
        wsrc[u] = W + (size_t)(f0 + row) * K + chunk * 8;
        int tr = t0 + row;
        if (tr > M - 1) tr = M - 1;
        xsrc[u] = X + (size_t)tr * g.ldx + chunk * 8;
    }
    auto stage = [&](int kt, int buf) {
        char *wl = smem + buf * 2 * TILE_BYTES + wave * 32 * 128;
        char *xl = wl + TILE_BYTES;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wsrc[u] + kt * BK), (lds_ptr_t)(wl + u * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(xsrc[u] + kt * BK), (lds_ptr_t)(xl + u * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
        const char *wl = smem + buf * 2 * TILE_BYTES;
        const char *xl = wl + TILE_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            vec8 wf_[4], xf_[4];
            const int c = h + 4 * s;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int wr = wf * 64 + i * 16 + r16;
                wf_[i] = *reinterpret_cast<const vec8 *>(wl + wr * 128 + ((c ^ (wr & 7)) << 4));
                const int xr = wt * 64 + i * 16 + r16;
                xf_[i] = *reinterpret_cast<const vec8 *>(xl + xr * 128 + ((c ^ (xr & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = E::mfma16(wf_[i], xf_[j], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // epilogue: acc[i][j][e] = out[token t0 + wt*64 + 16j + r16][feature f0 + wf*64 + 16i + 4h + e]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = t0 + wt * 64 + j * 16 + r16;
        if (t >= M) continue;
        size_t orow = (size_t)t;
        const float *pos = nullptr;
        if (EPI == EPI_PATCH) {  // GEMM row = frame*P + p  ->  token row frame*T + 1 + p, plus pos[1 + p]
            const int fr = t / g.P, p = t - fr * g.P;
            orow = (size_t)fr * g.T + 1 + p;
            pos = g.pos + (size_t)(1 + p) * g.N;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = f0 + wf * 64 + i * 16 + 4 * h;
            const float4 b4 = *reinterpret_cast<const float4 *>(g.bias + f);
            float v[4] = {acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w};
            if (EPI == EPI_STORE16 || EPI == EPI_GELU16 || EPI == EPI_QGELU16) {
                uint16_t o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float y = v[e];
                    if (EPI == EPI_GELU16) y = gelu_erf(y);
                    if (EPI == EPI_QGELU16) y = quick_gelu(y);
                    o[e] = E::from_float(y);
                }
                uint2 pk;
                __builtin_memcpy(&pk, o, 8);
                *reinterpret_cast<uint2 *>(g.out16 + orow * g.ldo + f) = pk;
            } else if (EPI == EPI_RESID32) {
                float4 *dst = reinterpret_cast<float4 *>(g.out32 + orow * g.ldo + f);
                float4 r = *dst;
                r.x += v[0];
                r.y += v[1];
                r.z += v[2];
                r.w += v[3];
                *dst = r;
            } else {  // EPI_PATCH
                const float4 p4 = *reinterpret_cast<const float4 *>(pos + f);
                *reinterpret_cast<float4 *>(g.out32 + orow * g.ldo + f) =
                    make_float4(v[0] + p4.x, v[1] + p4.y, v[2] + p4.z, v[3] + p4.w);
            }
        }
    }
}

template <int DT>
int launch(vm_ctx *ctx, const GemmArgs &g, int epi, hipStream_t st) {
    const int tiles = ((g.M + BM - 1) / BM) * (g.N / BN);
    const size_t lds = 4 * TILE_BYTES;
    const int cat = epi == EPI_PATCH ? VM_PROF_GEMM_PATCH
                    : epi == EPI_STORE16 ? VM_PROF_GEMM_QKV
                    : epi == EPI_RESID32 ? VM_PROF_GEMM_RESID : VM_PROF_GEMM_ACT;
    vm_prof_scope prof(ctx, cat, st);
#define GO(EPIV)                                                                                            \
    do {                                                                                                    \
        auto kern = gemm_kernel<DT, EPIV>;                                                                  \
        static bool attr_set = false;                                                                       \
        if (!attr_set) {                                                                                    \
            VM_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                            (int)lds));                                                     \
            attr_set = true;                                                                                \
        }                                                                                                   \
        kern<<<tiles, 256, lds, st>>>(g);                                                                   \
    } while (0)
    switch (epi) {
        case EPI_STORE16: GO(EPI_STORE16); break;
        case EPI_GELU16: GO(EPI_GELU16); break;
        case EPI_QGELU16: GO(EPI_QGELU16); break;
        case EPI_RESID32: GO(EPI_RESID32); break;
        case EPI_PATCH: GO(EPI_PATCH); break;
        default: return vm_fail(ctx, VM_ERR_INVALID, "bad epilogue %d", epi);
    }
#undef GO
    VM_LAUNCH_CHECK(ctx);
    return VM_OK;
}

}  // namespace

int vm_gemm(vm_ctx *ctx, int dtype, const GemmArgs &g, int epi, hipStream_t st) {
    if (g.M <= 0 || g.N % BN != 0 || g.K % BK != 0 || g.K <= 0)
        return vm_fail(ctx, VM_ERR_UNSUPPORTED, "gemm shape M=%d N=%d K=%d (need N%%128==0, K%%64==0)", g.M, g.N,
                       g.K);
    return dtype == VM_F16 ? launch<VM_F16>(ctx, g, epi, st) : launch<VM_BF16>(ctx, g, epi, st);
}
