// Vision encoder forward (ViT-B/16-224, CLIP-ViT-L/14-336 and anything of the same family):
// patch rows -> patch-embed GEMM (+bias +pos) -> [pre-LN] -> L x {LN1 -> QKV GEMM -> attention -> proj GEMM (+resid)
// -> LN2 -> FC1 GEMM (+act) -> FC2 GEMM (+resid)} -> final LN of the CLS row -> [projection] -> [L2 norm] -> 16 bit.
//
// This is the work the reference leaves to a remote model server: VLMExtractor._call_vlm_api
// (src/pipeline/vlm_extractor.py:130-185) and OpenAIEmbeddings.aembed_query (src/components/neo4j_handler.py:27-31,
// src/components/pre_llm_injector.py:207-221).  Residual stream fp32; every GEMM operand 16 bit; accumulation fp32.
#include "vm_internal.h"
#include "vm_kernels.h"

#include <type_traits>

// ---------------------------------------------------------------------------------------------------------
// HBM-bound kernels between the GEMMs
// ---------------------------------------------------------------------------------------------------------
namespace {

// Wave-wide sum in 6 DPP adds + one readlane (no LDS round trips): quad swaps, half-row and row mirrors give every
// lane its 16-lane row sum; row_bcast15 / row_bcast31 carry the row sums forward so lane 63 holds the total.
__device__ __forceinline__ float wave_sum(float v) {
    auto dpp_add = [](float x, auto ctrl, auto row_mask) {
        const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value,
                                                  decltype(row_mask)::value, 0xf, false);
        return x + __builtin_bit_cast(float, y);
    };
    using std::integral_constant;
    v = dpp_add(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xf>{});   // quad_perm [1,0,3,2]
    v = dpp_add(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xf>{});   // quad_perm [2,3,0,1]
    v = dpp_add(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xf>{});  // row_half_mirror
    v = dpp_add(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xf>{});  // row_mirror
    v = dpp_add(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});  // row_bcast15 -> rows 1, 3
    v = dpp_add(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});  // row_bcast31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Streaming (non-temporal) forms for rows that are touched once per pass and are far larger than L2 together.
typedef float f32x4_v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_v __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ float4 ld_f4(const float4 *p) {
    if (NT) {
        const f32x4_v v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_v *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *p;
}
template <bool NT>
__device__ __forceinline__ void st_f4(float4 *p, float4 v) {
    if (NT) __builtin_nontemporal_store(f32x4_v{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4_v *>(p));
    else *p = v;
}
template <bool NT>
__device__ __forceinline__ void st_u2(uint2 *p, uint2 v) {
    if (NT) __builtin_nontemporal_store(u32x2_v{v.x, v.y}, reinterpret_cast<u32x2_v *>(p));
    else *p = v;
}

template <int DT, bool NT = false>
__device__ __forceinline__ float4 load4_16(const uint16_t *p) {
    using E = vm_elem<DT>;
    uint2 v;
    if (NT) {
        const u32x2_v t = __builtin_nontemporal_load(reinterpret_cast<const u32x2_v *>(p));
        v = make_uint2(t.x, t.y);
    } else {
        v = *reinterpret_cast<const uint2 *>(p);
    }
    const uint16_t *e = reinterpret_cast<const uint16_t *>(&v);
    return make_float4(E::to_float(e[0]), E::to_float(e[1]), E::to_float(e[2]), E::to_float(e[3]));
}

// Residual add + LayerNorm, one wave per token row, H = 256 * VPL, lane owns 4-element chunks lane + 64*i.
//   x_new = (x32[row] + dA[row]) + dB[row] (the 16-bit outputs of the residual branches not yet folded into x32; either
//                                           may be null)
//   x32[row] = x_new                       only when write_x: the pass in front of the attention block folds BOTH pending
//                                           branch outputs (projection, FC2) into the fp32 stream; the pass in front of
//                                           the MLP only reads x32 + projection and leaves the fold to the next one
//                                           (22 instead of 24 bytes per element and layer: the kernel runs at the HBM
//                                           roofline, so only fewer bytes make it faster)
//   out16[row] = LN(x_new) * gamma + beta  (the next GEMM's A operand)
// Two-pass statistics in registers (mean, then mean of squared deviations), as the oracle computes them.
// The GEMMs therefore never read the residual: their epilogues are pure 16-bit stores.
template <int DT, int VPL, int RPW, bool NT>  // RPW rows per wave: all loads of both rows are in flight before the first use
__global__ void __launch_bounds__(256) resid_layernorm_kernel(float *__restrict__ x32,
                                                              const uint16_t *__restrict__ delta16,
                                                              const uint16_t *__restrict__ deltaB16, int write_x,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, float eps,
                                                              uint16_t *__restrict__ out16, int rows, int H,
                                                              int rstride) {  // row r lives at row r * rstride of every array
    using E = vm_elem<DT>;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    float4 v[RPW][VPL];
    float4 d[RPW][VPL];
    float4 e[RPW][VPL];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r < rows ? row0 + r : rows - 1;
        const float4 *xr = reinterpret_cast<const float4 *>(x32 + (size_t)row * rstride * H);
#pragma unroll
        for (int i = 0; i < VPL; ++i) v[r][i] = ld_f4<NT>(xr + lane + 64 * i);
        if (delta16) {
            const uint16_t *dr = delta16 + (size_t)row * rstride * H;
#pragma unroll
            for (int i = 0; i < VPL; ++i) d[r][i] = load4_16<VM_F16, NT>(dr + 4 * (lane + 64 * i));  // EPI_DELTA16: always fp16
        }
        if (deltaB16) {
            const uint16_t *er = deltaB16 + (size_t)row * rstride * H;
#pragma unroll
            for (int i = 0; i < VPL; ++i) e[r][i] = load4_16<VM_F16, NT>(er + 4 * (lane + 64 * i));
        }
    }
    float4 g4[VPL], b4[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        g4[i] = reinterpret_cast<const float4 *>(gamma)[lane + 64 * i];
        b4[i] = reinterpret_cast<const float4 *>(beta)[lane + 64 * i];
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r;
        if (row >= rows) break;
        if (delta16) {
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                v[r][i].x += d[r][i].x;
                v[r][i].y += d[r][i].y;
                v[r][i].z += d[r][i].z;
                v[r][i].w += d[r][i].w;
            }
        }
        if (deltaB16) {  // second, later branch output: added after the first, as the unfused sequence does
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                v[r][i].x += e[r][i].x;
                v[r][i].y += e[r][i].y;
                v[r][i].z += e[r][i].z;
                v[r][i].w += e[r][i].w;
            }
        }
        if (write_x > 1 ? row % write_x == 0 : write_x) {  // write_x = n > 1: only every n-th row's sum is read again
            float4 *xr = reinterpret_cast<float4 *>(x32 + (size_t)row * rstride * H);
#pragma unroll
            for (int i = 0; i < VPL; ++i) st_f4<NT>(xr + lane + 64 * i, v[r][i]);
        }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) sum += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
        const float mean = wave_sum(sum) / (float)H;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const float a = v[r][i].x - mean, b = v[r][i].y - mean, c = v[r][i].z - mean, e = v[r][i].w - mean;
            sq += (a * a + b * b) + (c * c + e * e);
        }
        const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            uint16_t o[4] = {E::from_float((v[r][i].x - mean) * rstd * g4[i].x + b4[i].x),
                             E::from_float((v[r][i].y - mean) * rstd * g4[i].y + b4[i].y),
                             E::from_float((v[r][i].z - mean) * rstd * g4[i].z + b4[i].z),
                             E::from_float((v[r][i].w - mean) * rstd * g4[i].w + b4[i].w)};
            uint2 pk;
            __builtin_memcpy(&pk, o, 8);
            st_u2<NT>(reinterpret_cast<uint2 *>(out16 + (size_t)row * rstride * H) + lane + 64 * i, pk);
        }
    }
}

// The same pass for co-residency with the persistent GEMM (vm_encode's two-stream mode): that kernel holds two waves of
// 224 VGPRs on every SIMD and all but 4 KiB of the LDS, so only a kernel of <= 64 VGPRs and no LDS is admitted beside
// it.  One row per wave, the 16-bit branch outputs kept packed until they are added, gamma / beta fetched (from L2)
// only after the statistics: same values, same additions in the same order, bit for bit.
template <int DT, int VPL, bool NT>
__global__ void __launch_bounds__(256, 8) resid_layernorm_lowreg_kernel(float *__restrict__ x32,
                                                                        const uint16_t *__restrict__ delta16,
                                                                        const uint16_t *__restrict__ deltaB16,
                                                                        int write_x, const float *__restrict__ gamma,
                                                                        const float *__restrict__ beta, float eps,
                                                                        uint16_t *__restrict__ out16, int rows, int H,
                                                                        int rstride) {
    using E = vm_elem<DT>;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const size_t base = (size_t)row * rstride * H;
    float4 v[VPL];
    const float4 *xr = reinterpret_cast<const float4 *>(x32 + base);
#pragma unroll
    for (int i = 0; i < VPL; ++i) v[i] = ld_f4<NT>(xr + lane + 64 * i);
    if (delta16) {
        float4 d[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) d[i] = load4_16<VM_F16, NT>(delta16 + base + 4 * (lane + 64 * i));
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            v[i].x += d[i].x;
            v[i].y += d[i].y;
            v[i].z += d[i].z;
            v[i].w += d[i].w;
        }
    }
    if (deltaB16) {
        float4 d[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) d[i] = load4_16<VM_F16, NT>(deltaB16 + base + 4 * (lane + 64 * i));
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            v[i].x += d[i].x;
            v[i].y += d[i].y;
            v[i].z += d[i].z;
            v[i].w += d[i].w;
        }
    }
    if (write_x > 1 ? row % write_x == 0 : write_x) {
        float4 *xw = reinterpret_cast<float4 *>(x32 + base);
#pragma unroll
        for (int i = 0; i < VPL; ++i) st_f4<NT>(xw + lane + 64 * i, v[i]);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum(sum) / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
        sq += (a * a + b * b) + (c * c + e * e);
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const float4 g4 = reinterpret_cast<const float4 *>(gamma)[lane + 64 * i];
        const float4 b4 = reinterpret_cast<const float4 *>(beta)[lane + 64 * i];
        uint16_t o[4] = {E::from_float((v[i].x - mean) * rstd * g4.x + b4.x),
                         E::from_float((v[i].y - mean) * rstd * g4.y + b4.y),
                         E::from_float((v[i].z - mean) * rstd * g4.z + b4.z),
                         E::from_float((v[i].w - mean) * rstd * g4.w + b4.w)};
        uint2 pk;
        __builtin_memcpy(&pk, o, 8);
        st_u2<NT>(reinterpret_cast<uint2 *>(out16 + base) + lane + 64 * i, pk);
    }
}

// Embedding assembly, one wave per token row: x32[frame*T + tok] = (tok ? patch16[frame*P + tok - 1] : cls) + pos[tok],
// followed by the optional pre-LayerNorm of CLIP (fp32 in, fp32 out).
template <int DT, int VPL>
__global__ void __launch_bounds__(256) embed_kernel(const uint16_t *__restrict__ patch16, const float *__restrict__ cls,
                                                    const float *__restrict__ pos, const float *__restrict__ pre_g,
                                                    const float *__restrict__ pre_b, float eps, int pre_ln,
                                                    float *__restrict__ x32, int rows, int T, int H) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int frame = row / T, tok = row - frame * T;
    float4 v[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        const float4 p4 = reinterpret_cast<const float4 *>(pos + (size_t)tok * H)[c];
        float4 e4;
        if (tok == 0)
            e4 = reinterpret_cast<const float4 *>(cls)[c];
        else
            e4 = load4_16<VM_F16>(patch16 + ((size_t)frame * (T - 1) + tok - 1) * H + 4 * c);  // EPI_DELTA16 rows
        v[i] = make_float4(e4.x + p4.x, e4.y + p4.y, e4.z + p4.z, e4.w + p4.w);
    }
    if (pre_ln) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        const float mean = wave_sum(sum) / (float)H;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            sq += (a * a + b * b) + (c * c + d * d);
        }
        const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const float4 g4 = reinterpret_cast<const float4 *>(pre_g)[lane + 64 * i];
            const float4 b4 = reinterpret_cast<const float4 *>(pre_b)[lane + 64 * i];
            v[i] = make_float4((v[i].x - mean) * rstd * g4.x + b4.x, (v[i].y - mean) * rstd * g4.y + b4.y,
                               (v[i].z - mean) * rstd * g4.z + b4.z, (v[i].w - mean) * rstd * g4.w + b4.w);
        }
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) reinterpret_cast<float4 *>(x32 + (size_t)row * H)[lane + 64 * i] = v[i];
}

// One block per frame: CLS row = x32 + delta16 (the last FC2 output), final LayerNorm (fp32), optional projection
// W[proj_dim, H] (16-bit weights, 16-bit rounded input, fp32 accumulate), optional L2 normalisation, cast to 16 bit.
template <int DT>
__global__ void __launch_bounds__(256) pool_kernel(const float *__restrict__ x, const uint16_t *__restrict__ delta16,
                                                   const uint16_t *__restrict__ deltaB16,
                                                   const float *__restrict__ gamma, const float *__restrict__ beta,
                                                   float eps, const uint16_t *__restrict__ proj_w, int proj_dim, int l2,
                                                   uint16_t *__restrict__ out, int T, int H) {
    using E = vm_elem<DT>;
    extern __shared__ __attribute__((aligned(16))) float sh[];  // [H] row, then [out_dim] result
    __shared__ float red[8];
    float *y = sh;
    float *res = sh + H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *row = x + (size_t)blockIdx.x * T * H;
    const uint16_t *drow = delta16 + (size_t)blockIdx.x * T * H;
    const uint16_t *erow = deltaB16 + (size_t)blockIdx.x * T * H;
    auto block_sum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    };
    float s = 0.f;
    for (int i = tid; i < H; i += 256) {
        // the last layer's two branch outputs (EPI_DELTA16: always fp16), in branch order
        const float v = (row[i] + vm_elem<VM_F16>::to_float(drow[i])) + vm_elem<VM_F16>::to_float(erow[i]);
        y[i] = v;
        s += v;
    }
    const float mean = block_sum(s) / (float)H;
    float sq = 0.f;
    for (int i = tid; i < H; i += 256) {
        const float d = y[i] - mean;
        sq += d * d;
    }
    const float rstd = rsqrtf(block_sum(sq) / (float)H + eps);
    for (int i = tid; i < H; i += 256) y[i] = (y[i] - mean) * rstd * gamma[i] + beta[i];
    __syncthreads();
    int od = H;
    const float *src = y;
    if (proj_dim > 0) {
        od = proj_dim;
        for (int o = wave; o < proj_dim; o += 4) {  // one wave per output row
            const uint16_t *w = proj_w + (size_t)o * H;
            float acc = 0.f;
            for (int i = lane; i < H; i += 64) acc += E::to_float(E::from_float(y[i])) * E::to_float(w[i]);
            acc = wave_sum(acc);
            if (lane == 0) res[o] = acc;
        }
        __syncthreads();
        src = res;
    }
    float nn = 0.f;
    for (int i = tid; i < od; i += 256) nn += src[i] * src[i];
    float inv = 1.f;
    if (l2) {
        const float nrm = sqrtf(block_sum(nn));
        inv = 1.f / fmaxf(nrm, 1e-12f);
    }
    for (int i = tid; i < od; i += 256) out[(size_t)blockIdx.x * od + i] = E::from_float(src[i] * inv);
}

}  // namespace

#define VM_VPL_SWITCH(H, CALL)                                                                     \
    switch ((H) / 256) {                                                                           \
        case 1: CALL(1); break;                                                                    \
        case 2: CALL(2); break;                                                                    \
        case 3: CALL(3); break;                                                                    \
        case 4: CALL(4); break;                                                                    \
        default: return vm_fail(ctx, VM_ERR_UNSUPPORTED, "row width %d (need 256..1024, %%256)", H); \
    }

int vm_resid_layernorm(vm_ctx *ctx, int dtype, float *x32, const uint16_t *delta16, const uint16_t *deltaB16,
                       int write_x, const float *gamma, const float *beta, float eps, uint16_t *out16, int rows, int H,
                       hipStream_t st, int rstride, int lowreg) {
    constexpr int RPW = 2;
    const int blocks = (rows + 4 * RPW - 1) / (4 * RPW);
    if (H % 256 != 0) return vm_fail(ctx, VM_ERR_UNSUPPORTED, "row width %d", H);
    vm_prof_scope prof(ctx, VM_PROF_LAYERNORM, st);
    static const int lowreg_env = (int)VM_DEV_ENV("LN_LOWREG", 1);   // developer A/B: 0 = the ordinary kernel on two streams as well
    // (round 4, three alternating pairs: ViT-B/16 26.78-26.84 k against 27.23-27.25 k frames/s with this one, CLIP-L 2,691 / 2,804)
    if (lowreg != 0 && lowreg_env != 0) {   // two-stream mode: the variant that fits beside the persistent GEMM (always the streaming policy)
        const int blk = (rows + 3) / 4;
#define RLNL(V)                                                                                                          \
    if (dtype == VM_F16)                                                                                                 \
        resid_layernorm_lowreg_kernel<VM_F16, V, true><<<blk, 256, 0, st>>>(x32, delta16, deltaB16, write_x, gamma, beta, eps, out16, rows, H, rstride); \
    else                                                                                                                 \
        resid_layernorm_lowreg_kernel<VM_BF16, V, true><<<blk, 256, 0, st>>>(x32, delta16, deltaB16, write_x, gamma, beta, eps, out16, rows, H, rstride)
        VM_VPL_SWITCH(H, RLNL)
#undef RLNL
        VM_LAUNCH_CHECK(ctx);
        return VM_OK;
    }
    // rows that together exceed the 32 MiB of L2 several times over stream through with the non-temporal policy
    static const int nt_env = (int)VM_DEV_ENV("LN_NT", 1);
    const bool nt = nt_env && (size_t)rows * H * 4 > ((size_t)64 << 20);
#define RLN16(V) resid_layernorm_kernel<VM_F16, V, RPW, false><<<blocks, 256, 0, st>>>(x32, delta16, deltaB16, write_x, gamma, beta, eps, out16, rows, H, rstride)
#define RLNB16(V) resid_layernorm_kernel<VM_BF16, V, RPW, false><<<blocks, 256, 0, st>>>(x32, delta16, deltaB16, write_x, gamma, beta, eps, out16, rows, H, rstride)
#define RLN16N(V) resid_layernorm_kernel<VM_F16, V, RPW, true><<<blocks, 256, 0, st>>>(x32, delta16, deltaB16, write_x, gamma, beta, eps, out16, rows, H, rstride)
#define RLNB16N(V) resid_layernorm_kernel<VM_BF16, V, RPW, true><<<blocks, 256, 0, st>>>(x32, delta16, deltaB16, write_x, gamma, beta, eps, out16, rows, H, rstride)
    if (dtype == VM_F16) {
        if (nt) { VM_VPL_SWITCH(H, RLN16N) } else { VM_VPL_SWITCH(H, RLN16) }
    } else {
        if (nt) { VM_VPL_SWITCH(H, RLNB16N) } else { VM_VPL_SWITCH(H, RLNB16) }
    }
#undef RLN16
#undef RLNB16
#undef RLN16N
#undef RLNB16N
    VM_LAUNCH_CHECK(ctx);
    return VM_OK;
}

int vm_embed(vm_ctx *ctx, int dtype, const uint16_t *patch16, const float *cls, const float *pos, const float *pre_g,
             const float *pre_b, float eps, int pre_ln, float *x32, int B, int T, int H, hipStream_t st) {
    const int rows = B * T, blocks = (rows + 3) / 4;
    if (H % 256 != 0) return vm_fail(ctx, VM_ERR_UNSUPPORTED, "row width %d", H);
    vm_prof_scope prof(ctx, VM_PROF_LAYERNORM, st);
#define EMB16(V) embed_kernel<VM_F16, V><<<blocks, 256, 0, st>>>(patch16, cls, pos, pre_g, pre_b, eps, pre_ln, x32, rows, T, H)
#define EMBB16(V) embed_kernel<VM_BF16, V><<<blocks, 256, 0, st>>>(patch16, cls, pos, pre_g, pre_b, eps, pre_ln, x32, rows, T, H)
    if (dtype == VM_F16) {
        VM_VPL_SWITCH(H, EMB16)
    } else {
        VM_VPL_SWITCH(H, EMBB16)
    }
#undef EMB16
#undef EMBB16
    VM_LAUNCH_CHECK(ctx);
    return VM_OK;
}

int vm_pool(vm_ctx *ctx, int dtype, const float *x, const uint16_t *delta16, const uint16_t *deltaB16,
            const float *gamma, const float *beta, float eps, const uint16_t *proj_w, int proj_dim, int l2, uint16_t *out,
            int B, int T, int H, hipStream_t st) {
    const size_t lds = (size_t)(H + (proj_dim > 0 ? proj_dim : 0)) * 4;
    vm_prof_scope prof(ctx, VM_PROF_POOL, st);
    if (dtype == VM_F16)
        pool_kernel<VM_F16><<<B, 256, lds, st>>>(x, delta16, deltaB16, gamma, beta, eps, proj_w, proj_dim, l2, out, T, H);
    else
        pool_kernel<VM_BF16><<<B, 256, lds, st>>>(x, delta16, deltaB16, gamma, beta, eps, proj_w, proj_dim, l2, out, T, H);
    VM_LAUNCH_CHECK(ctx);
    return VM_OK;
}

// ---------------------------------------------------------------------------------------------------------
// handle + orchestration
// ---------------------------------------------------------------------------------------------------------
struct LayerW {
    float *ln1_g, *ln1_b, *qkv_b, *proj_b, *ln2_g, *ln2_b, *fc1_b, *fc2_b;
    uint16_t *qkv_w, *proj_w, *fc1_w, *fc2_w;
};
struct vm_encoder {
    vm_ctx *ctx;
    vm_encoder_desc d;
    int tokens, patches, patch_k, out_dim;
    char *blob;  // one device allocation holding every weight
    uint16_t *patch_w, *proj_w;
    float *patch_b, *cls, *pos, *pre_g, *pre_b, *ln_g, *ln_b;
    LayerW *layers;
    int micro_batch;   // VM_ENC_OPT_MICRO_BATCH: frames per pass, 0 = auto
    int cls_last;      // VM_ENC_OPT_LAST_LAYER (default 3): see vm_encode's last layer
    // VM_ENC_OPT_SCHEDULE.  Two-stream schedule: consecutive micro-batch passes of one vm_encode call alternate between
    // two internal streams, so that the bandwidth-bound LayerNorms of one pass (low-register build) run beside the
    // matrix-bound GEMMs of the other; see vm_encode.  AUTO = two streams unless per-kernel timing is on.
    int schedule;
    hipStream_t side[2];
    hipEvent_t ev_fork, ev_join[2], ev_phase;   // ev_phase: developer experiment (start offset of the second stream)
};

static int round_up(int x, int a) { return (x + a - 1) / a * a; }

extern "C" int vm_encoder_create(vm_ctx *ctx, const vm_encoder_desc *desc, const void *const *wp, int n_weights,
                                 vm_encoder **out) {
    if (!ctx || !desc || !wp || !out) return VM_ERR_INVALID;
    const vm_encoder_desc &d = *desc;
    if (d.hidden % 256 != 0 || d.hidden > 1024 || d.heads * 64 != d.hidden)
        return vm_fail(ctx, VM_ERR_UNSUPPORTED, "hidden=%d heads=%d: need hidden %% 256 == 0, <= 1024, head dim 64",
                       d.hidden, d.heads);
    if (d.mlp % 128 != 0 || d.image % d.patch != 0 || d.layers <= 0 || d.image % 8 != 0)
        return vm_fail(ctx, VM_ERR_UNSUPPORTED, "bad mlp/image/patch/layers");
    if (d.dtype != VM_F16 && d.dtype != VM_BF16) return vm_fail(ctx, VM_ERR_INVALID, "bad dtype");
    if (d.proj_dim < 0 || d.proj_dim % 8 != 0) return vm_fail(ctx, VM_ERR_UNSUPPORTED, "proj_dim %d", d.proj_dim);
    if (n_weights != 9 + 12 * d.layers)
        return vm_fail(ctx, VM_ERR_INVALID, "expected %d weight pointers, got %d", 9 + 12 * d.layers, n_weights);
    VM_HIP(ctx, hipSetDevice(ctx->device));
    vm_encoder *e = new vm_encoder();
    memset(e, 0, sizeof(*e));
    e->ctx = ctx;
    e->d = d;
    const int g = d.image / d.patch;
    e->patches = g * g;
    e->tokens = e->patches + 1;
    e->patch_k = round_up(3 * d.patch * d.patch, 64);
    e->out_dim = d.proj_dim ? d.proj_dim : d.hidden;
    if (e->tokens > 592) {
        const int tokens = e->tokens;
        delete e;
        return vm_fail(ctx, VM_ERR_UNSUPPORTED, "%d tokens per frame > 592", tokens);
    }
    e->micro_batch = 0;
    e->cls_last = 3;
    e->schedule = VM_SCHED_AUTO;
    {   // the two internal streams of the two-stream schedule (a few hundred bytes of driver state when never used)
        hipError_t he = hipSuccess;
        for (int i = 0; i < 2 && he == hipSuccess; ++i) {
            he = hipStreamCreateWithFlags(&e->side[i], hipStreamNonBlocking);
            if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_join[i], hipEventDisableTiming);
        }
        if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_phase, hipEventDisableTiming);
        if (he != hipSuccess) {
            vm_encoder_destroy(e);   // destroys whatever was created so far
            return vm_fail(ctx, VM_ERR_HIP, "encoder streams: %s", hipGetErrorString(he));
        }
    }
    const size_t H = d.hidden, M = d.mlp;
    // byte sizes in header order
    auto a256 = [](size_t b) { return vm_align_up(b, 256); };
    size_t total = 0;
    auto take = [&](size_t bytes) {
        size_t off = total;
        total += a256(bytes);
        return off;
    };
    const size_t o_patch_w = take(H * e->patch_k * 2), o_patch_b = take(H * 4), o_cls = take(H * 4),
                 o_pos = take((size_t)e->tokens * H * 4), o_pre_g = take(H * 4), o_pre_b = take(H * 4),
                 o_ln_g = take(H * 4), o_ln_b = take(H * 4), o_proj = take((size_t)(d.proj_dim ? d.proj_dim : 0) * H * 2);
    struct LO {
        size_t o[12];
    };
    LO *lo = new LO[d.layers];
    const size_t lsz[12] = {H * 4, H * 4, 3 * H * H * 2, 3 * H * 4, H * H * 2, H * 4,
                            H * 4, H * 4, M * H * 2,     M * 4,     H * M * 2, H * 4};
    for (int l = 0; l < d.layers; ++l)
        for (int i = 0; i < 12; ++i) lo[l].o[i] = take(lsz[i]);
    hipError_t er = hipMalloc((void **)&e->blob, total);
    if (er != hipSuccess) {
        delete[] lo;
        delete e;
        return vm_fail(ctx, VM_ERR_NOMEM, "encoder weights: hipMalloc(%zu) failed", total);
    }
    auto copy = [&](size_t off, const void *src, size_t bytes) -> hipError_t {
        if (bytes == 0) return hipSuccess;
        if (!src) return hipErrorInvalidValue;
        return hipMemcpy(e->blob + off, src, bytes, hipMemcpyDeviceToDevice);
    };
    er = copy(o_patch_w, wp[0], H * e->patch_k * 2);
    if (er == hipSuccess) er = d.patch_bias ? copy(o_patch_b, wp[1], H * 4) : hipMemset(e->blob + o_patch_b, 0, H * 4);
    if (er == hipSuccess) er = copy(o_cls, wp[2], H * 4);
    if (er == hipSuccess) er = copy(o_pos, wp[3], (size_t)e->tokens * H * 4);
    if (er == hipSuccess && d.pre_ln) er = copy(o_pre_g, wp[4], H * 4);
    if (er == hipSuccess && d.pre_ln) er = copy(o_pre_b, wp[5], H * 4);
    if (er == hipSuccess) er = copy(o_ln_g, wp[6], H * 4);
    if (er == hipSuccess) er = copy(o_ln_b, wp[7], H * 4);
    if (er == hipSuccess && d.proj_dim) er = copy(o_proj, wp[8], (size_t)d.proj_dim * H * 2);
    e->layers = new LayerW[d.layers];
    for (int l = 0; l < d.layers && er == hipSuccess; ++l) {
        for (int i = 0; i < 12 && er == hipSuccess; ++i) er = copy(lo[l].o[i], wp[9 + 12 * l + i], lsz[i]);
        LayerW &w = e->layers[l];
        char *b = e->blob;
        w.ln1_g = (float *)(b + lo[l].o[0]);
        w.ln1_b = (float *)(b + lo[l].o[1]);
        w.qkv_w = (uint16_t *)(b + lo[l].o[2]);
        w.qkv_b = (float *)(b + lo[l].o[3]);
        w.proj_w = (uint16_t *)(b + lo[l].o[4]);
        w.proj_b = (float *)(b + lo[l].o[5]);
        w.ln2_g = (float *)(b + lo[l].o[6]);
        w.ln2_b = (float *)(b + lo[l].o[7]);
        w.fc1_w = (uint16_t *)(b + lo[l].o[8]);
        w.fc1_b = (float *)(b + lo[l].o[9]);
        w.fc2_w = (uint16_t *)(b + lo[l].o[10]);
        w.fc2_b = (float *)(b + lo[l].o[11]);
    }
    delete[] lo;
    if (er != hipSuccess) {
        vm_encoder_destroy(e);
        return vm_fail(ctx, VM_ERR_HIP, "encoder weight copy failed: %s (null or short weight pointer?)",
                       hipGetErrorString(er));
    }
    char *b = e->blob;
    e->patch_w = (uint16_t *)(b + o_patch_w);
    e->patch_b = (float *)(b + o_patch_b);
    e->cls = (float *)(b + o_cls);
    e->pos = (float *)(b + o_pos);
    e->pre_g = (float *)(b + o_pre_g);
    e->pre_b = (float *)(b + o_pre_b);
    e->ln_g = (float *)(b + o_ln_g);
    e->ln_b = (float *)(b + o_ln_b);
    e->proj_w = d.proj_dim ? (uint16_t *)(b + o_proj) : nullptr;
    *out = e;
    return VM_OK;
}

extern "C" void vm_encoder_destroy(vm_encoder *e) {
    if (!e) return;
    for (int i = 0; i < 2; ++i) {
        if (e->side[i]) (void)hipStreamDestroy(e->side[i]);
        if (e->ev_join[i]) (void)hipEventDestroy(e->ev_join[i]);
    }
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_phase) (void)hipEventDestroy(e->ev_phase);
    if (e->blob) (void)hipFree(e->blob);
    delete[] e->layers;
    delete e;
}

extern "C" int vm_encoder_set_option(vm_encoder *e, int option, int value) {
    if (!e) return VM_ERR_INVALID;
    switch (option) {
        case VM_ENC_OPT_SCHEDULE:
            if (value != VM_SCHED_AUTO && value != VM_SCHED_ONE_STREAM && value != VM_SCHED_TWO_STREAMS)
                return vm_fail(e->ctx, VM_ERR_INVALID, "VM_ENC_OPT_SCHEDULE: %d", value);
            e->schedule = value;
            return VM_OK;
        case VM_ENC_OPT_MICRO_BATCH:
            if (value < 0) return vm_fail(e->ctx, VM_ERR_INVALID, "VM_ENC_OPT_MICRO_BATCH: %d", value);
            e->micro_batch = value;
            return VM_OK;
        case VM_ENC_OPT_LAST_LAYER:
            if (value != 0 && value != 1 && value != 3)
                return vm_fail(e->ctx, VM_ERR_INVALID, "VM_ENC_OPT_LAST_LAYER: %d (0, 1 or 3)", value);
            e->cls_last = value;
            return VM_OK;
        default:
            return vm_fail(e->ctx, VM_ERR_INVALID, "unknown encoder option %d", option);
    }
}
extern "C" int vm_encoder_get_option(const vm_encoder *e, int option) {
    if (!e) return VM_ERR_INVALID;
    switch (option) {
        case VM_ENC_OPT_SCHEDULE: return e->schedule;
        case VM_ENC_OPT_MICRO_BATCH: return e->micro_batch;
        case VM_ENC_OPT_LAST_LAYER: return e->cls_last;
        default: return VM_ERR_INVALID;
    }
}

extern "C" int vm_encoder_tokens(const vm_encoder *e) { return e ? e->tokens : 0; }
extern "C" int vm_encoder_patch_k(const vm_encoder *e) { return e ? e->patch_k : 0; }
extern "C" int vm_encoder_out_dim(const vm_encoder *e) { return e ? e->out_dim : 0; }

static int micro_batch_of(const vm_encoder *e, int B) {
    // Frames per pass.  The GEMMs use 256 x 256 tiles, one per CU per round: pick a batch whose token rows fill a
    // whole number of rounds for the narrowest GEMM (N = hidden): ceil(mb*T/256) * (hidden/256) = rounds * ~CUs.
    // ViT-B/16 (T=197, hidden 768): 85 row tiles x 3 = 255 tiles per round -> 110 frames per round;
    // CLIP-L/14-336 (T=577, hidden 1024): 64 x 4 = 256 tiles -> 28 frames per round.
    int mb = e->micro_batch;
    if (mb <= 0) {
        // 8 rounds of tiles per launch for the narrowest GEMM: fewer, longer launches amortise the ~6 us of ramp and
        // tail each of the ~85 kernels of a pass pays (measured: 4 rounds +3 % over 1 round; 8 rounds - the bench's 880
        // frames in ONE pass - another +1.8 %, 24.4 k -> 24.9 k frames/s on one box); 3.2 GB of workspace is nothing
        // next to 288 GB of HBM
        const int col_tiles = e->d.hidden / 256;
        const int row_tiles = 8 * (e->ctx->num_cus / col_tiles);
        mb = row_tiles * 256 / e->tokens;
        if (mb < 1) mb = 1;
        // Sequences past 208 tokens run ONE attention workgroup per (frame, head) and CU (attention.hip, long kernel):
        // mb * heads items in rounds of num_cus.  Give up < 3 % of the batch when that turns a nearly empty last round
        // into none (CLIP-L/14-336: 113 frames x 16 heads = 7.06 rounds -> 112 frames = 7 rounds; attention -12 %).
        if (e->tokens > 208) {
            const int items = mb * e->d.heads, cus = e->ctx->num_cus;
            const int full = items / cus * cus;
            if (full > 0 && full % e->d.heads == 0 && (items - full) * 32 < items) mb = full / e->d.heads;
        }
    }
    return B < mb ? B : mb;
}

struct Ws {
    float *x32;                            // fp32 residual stream [rows, H]
    uint16_t *a16, *d16, *e16, *qkv16, *mlp16;   // LN out / attention ctx, projection out (+ patch rows), FC2 out, QKV, MLP hidden
    size_t bytes;
};
static Ws carve(const vm_encoder *e, int mb, void *base) {
    const size_t rows = (size_t)mb * e->tokens, H = e->d.hidden, M = e->d.mlp;
    Ws w;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += vm_align_up(bytes, 256);
        return (char *)base + o;
    };
    w.x32 = (float *)take(rows * H * 4);
    w.a16 = (uint16_t *)take(rows * H * 2);
    w.d16 = (uint16_t *)take(rows * H * 2);
    w.e16 = (uint16_t *)take(rows * H * 2);
    w.qkv16 = (uint16_t *)take(rows * 3 * H * 2);
    w.mlp16 = (uint16_t *)take(rows * M * 2);
    w.bytes = off;
    return w;
}

extern "C" size_t vm_encode_workspace_bytes(const vm_encoder *e, int B) {
    if (!e || B <= 0) return 0;
    const int mb = micro_batch_of(e, B);
    const size_t one = carve(e, mb, nullptr).bytes;
    // the two-stream schedule keeps one workspace per stream (calls of more than one pass)
    return (e->schedule != VM_SCHED_ONE_STREAM && B > mb) ? 2 * one : one;
}

extern "C" int vm_encode_micro_batch(const vm_encoder *e, int B) { return e && B > 0 ? micro_batch_of(e, B) : 0; }

extern "C" int vm_encode(vm_encoder *e, const void *patches, int B, void *out_emb, int l2_normalise,
                         void *workspace, size_t workspace_bytes, void *stream) {
    if (!e) return VM_ERR_INVALID;
    vm_ctx *ctx = e->ctx;
    if (!patches || !out_emb || B <= 0) return vm_fail(ctx, VM_ERR_INVALID, "vm_encode: bad arguments");
    const int mb = micro_batch_of(e, B);
    const Ws ws0 = carve(e, mb, workspace);
    if (!workspace || workspace_bytes < ws0.bytes)
        return vm_fail(ctx, VM_ERR_NOMEM, "vm_encode: workspace %zu < %zu", workspace_bytes, ws0.bytes);
    if (((uintptr_t)workspace & 255) || ((uintptr_t)patches & 15) || ((uintptr_t)out_emb & 15))
        return vm_fail(ctx, VM_ERR_INVALID, "vm_encode: workspace must be 256-byte, tensors 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const vm_encoder_desc &d = e->d;
    const int H = d.hidden, T = e->tokens, P = e->patches, dt = d.dtype;
    const int act_epi = d.act == VM_ACT_QUICK_GELU ? EPI_QGELU16 : EPI_GELU16;
    // VM_ENC_OPT_LAST_LAYER: bit 0 = projection / LN2 / MLP of the last layer on the CLS rows only, bit 1 = also only
    // the CLS rows' queries and query tile in its attention; 0 = everything on every row (same embeddings, tests)
    const int cls_env = e->cls_last;
    const bool cls_only = (cls_env & 1) != 0;

    // Two-stream schedule (VM_ENC_OPT_SCHEDULE; the default for calls of two or more passes).  Consecutive micro-batch
    // passes of a call alternate between two internal streams (fork behind the caller's stream, join before returning;
    // one workspace per stream) and the residual + LayerNorm passes use their low-register build - 58 VGPRs, no LDS:
    // the only kernel of the encoder that is admitted NEXT TO the persistent GEMM, which leaves 64 VGPRs per SIMD and
    // 4 KiB of LDS free.  Measured (alternating A/B, embeddings bit-identical): ViT-B/16 fp16
    // 25.7 k -> 26.8 k frames/s (+4.3 %), CLIP-L/14-336 bf16 2,615 -> 2,703 (+3.3 %); with the ordinary LayerNorm on
    // two streams +-0; with every matrix kernel of both passes on ONE stream and only the LayerNorms on a side stream
    // between events (tools/experiments/encoder_kernel_granularity_pipeline.patch) -1.3 %: the GEMM launches stretch
    // by more than the LayerNorm time they cover.  With two streams a kernel's HIP-event (and rocprofv3) duration
    // includes the time it waits for the other stream's GEMM to leave the CUs, so VM_SCHED_AUTO falls back to ONE
    // stream while per-kernel timing is enabled on the context (vm_profile_enable > 0): same embeddings, clean timings.
    struct Pass {
        int b0, nb, rows;
        Ws ws;
        const uint16_t *pend_proj, *pend_fc2;   // branch outputs not yet folded into x32 (previous layer's)
    };
    const bool timing = ctx->prof_ev != nullptr;
    const bool dual = B > mb && workspace_bytes >= 2 * ws0.bytes &&
                      (e->schedule == VM_SCHED_TWO_STREAMS || (e->schedule == VM_SCHED_AUTO && !timing));
    const Ws ws1 = dual ? carve(e, mb, (char *)workspace + ws0.bytes) : ws0;
    hipStream_t st0 = st;
    int rc = VM_OK;
    if (dual) {   // fork: both internal streams start behind everything queued on the caller's stream
        VM_HIP(ctx, hipEventRecord(e->ev_fork, st0));
        for (int i = 0; i < 2; ++i) VM_HIP(ctx, hipStreamWaitEvent(e->side[i], e->ev_fork, 0));
    }
    GemmArgs g;
    int g_head_major = 0, g_hm_rows = 0, g_hm_stride = 0;
    auto gemm16 = [&](const uint16_t *X, int ldx, const uint16_t *W, const float *bias, uint16_t *out, int M, int N,
                      int K, int epi, int cat, int ldo = 0) {
        memset(&g, 0, sizeof(g));
        g.X = X; g.W = W; g.bias = bias; g.out16 = out;
        g.M = M; g.N = N; g.K = K; g.ldx = ldx; g.ldo = ldo ? ldo : N; g.prof_cat = cat; g.head_major = g_head_major;
        g.hm_rows = g_hm_rows; g.hm_stride = g_hm_stride;
        return vm_gemm(ctx, dt, g, epi, st);
    };
    // ---- the stages of one pass ----
    auto embed = [&](Pass &p) -> int {
        // patch embedding: [nb*P, patch_k] x [H, patch_k]^T (+bias) -> 16-bit rows; then x32 = rows + pos (+cls) [+pre-LN]
        int r = gemm16((const uint16_t *)patches + (size_t)p.b0 * P * e->patch_k, e->patch_k, e->patch_w, e->patch_b,
                       p.ws.d16, p.nb * P, H, e->patch_k, EPI_DELTA16, VM_PROF_GEMM_PATCH);
        if (r != VM_OK) return r;
        p.pend_proj = p.pend_fc2 = nullptr;
        return vm_embed(ctx, dt, p.ws.d16, e->cls, e->pos, e->pre_g, e->pre_b, d.ln_eps, d.pre_ln, p.ws.x32, p.nb, T, H, st);
    };
    auto ln1 = [&](Pass &p, int l) -> int {
        // x32 += proj(l-1) + fc2(l-1), written back once; a16 = LN1(x32)
        // (last layer: only the CLS rows' folded sums are read again - by LN2 and the pool - so only they are written)
        const LayerW &w = e->layers[l];
        const int fold = p.pend_proj == nullptr ? 0 : (l == d.layers - 1 && cls_only && T > 1 ? T : 1);
        return vm_resid_layernorm(ctx, dt, p.ws.x32, p.pend_proj, p.pend_fc2, fold, w.ln1_g, w.ln1_b, d.ln_eps, p.ws.a16,
                                  p.rows, H, st, 1, dual ? 1 : 0);
    };
    auto attn_block = [&](Pass &p, int l) -> int {   // QKV, attention, projection
        const LayerW &w = e->layers[l];
        const Ws &ws = p.ws;
        const int rows = p.rows, nb = p.nb;
        const bool last_cls = l == d.layers - 1 && cls_only;
        int r;
        g_head_major = 1;  // q/k/v of one head as contiguous [rows, 64] blocks: attention streams whole KiB
        if (last_cls && (cls_env & 2)) {
            // last layer: keys and values of every row, but only the CLS rows' queries (see below): the K / V
            // weight rows [H, 3H) write the k and v blocks, then a GEMM over the nb CLS rows (row stride T) writes
            // each head's query into row b*T of its q block
            r = gemm16(ws.a16, H, w.qkv_w + (size_t)H * H, w.qkv_b + H, ws.qkv16 + (size_t)d.heads * rows * 64, rows,
                       2 * H, H, EPI_STORE16, VM_PROF_GEMM_QKV);
            if (r == VM_OK) {
                g_hm_rows = rows;
                g_hm_stride = T;
                r = gemm16(ws.a16, T * H, w.qkv_w, w.qkv_b, ws.qkv16, nb, H, H, EPI_STORE16, VM_PROF_GEMM_CLS);
                g_hm_rows = g_hm_stride = 0;
            }
        } else {
            r = gemm16(ws.a16, H, w.qkv_w, w.qkv_b, ws.qkv16, rows, 3 * H, H, EPI_STORE16, VM_PROF_GEMM_QKV);
        }
        g_head_major = 0;
        if (r != VM_OK) return r;
        if ((r = vm_attention(ctx, dt, ws.qkv16, ws.a16, nb, T, d.heads, st, last_cls && (cls_env & 2) ? 1 : 0)) != VM_OK)
            return r;
        if (last_cls) {
            // LAST layer: the embedding is pooled from the CLS row alone (vm_pool), and behind the attention every
            // row depends only on itself - so projection, LN2, FC1 and FC2 run on the nb CLS rows, addressed in
            // place with a row stride of T rows (GEMM ldx / ldo, LN rstride).  The other rows' branch outputs were
            // never read by anything; the CLS rows get the same values bit for bit (every GEMM tiling accumulates
            // an output in the same MFMA order).  6.2 % of ViT-B/16's FLOPs, 3.5 % of CLIP-L/14-336's.
            const int TH = T * H;
            return gemm16(ws.a16, TH, w.proj_w, w.proj_b, ws.d16, nb, H, H, EPI_DELTA16, VM_PROF_GEMM_CLS, TH);
        }
        return gemm16(ws.a16, H, w.proj_w, w.proj_b, ws.d16, rows, H, H, EPI_DELTA16, VM_PROF_GEMM_RESID);
    };
    auto ln2 = [&](Pass &p, int l) -> int {
        // a16 = LN2(x32 + proj(l)); x32 itself is NOT rewritten here: the next LN1 (or the pool) folds both
        const LayerW &w = e->layers[l];
        const bool last_cls = l == d.layers - 1 && cls_only;
        return vm_resid_layernorm(ctx, dt, p.ws.x32, p.ws.d16, nullptr, 0, w.ln2_g, w.ln2_b, d.ln_eps, p.ws.a16,
                                  last_cls ? p.nb : p.rows, H, st, last_cls ? T : 1, dual && !last_cls ? 1 : 0);
    };
    auto mlp_block = [&](Pass &p, int l) -> int {   // FC1 (+activation), FC2
        const LayerW &w = e->layers[l];
        const Ws &ws = p.ws;
        const bool last_cls = l == d.layers - 1 && cls_only;
        int r;
        if (last_cls) {
            const int TH = T * H;
            if ((r = gemm16(ws.a16, TH, w.fc1_w, w.fc1_b, ws.mlp16, p.nb, d.mlp, H, act_epi, VM_PROF_GEMM_CLS)) != VM_OK) return r;
            r = gemm16(ws.mlp16, d.mlp, w.fc2_w, w.fc2_b, ws.e16, p.nb, H, d.mlp, EPI_DELTA16, VM_PROF_GEMM_CLS, TH);
        } else {
            if ((r = gemm16(ws.a16, H, w.fc1_w, w.fc1_b, ws.mlp16, p.rows, d.mlp, H, act_epi, VM_PROF_GEMM_ACT)) != VM_OK) return r;
            r = gemm16(ws.mlp16, d.mlp, w.fc2_w, w.fc2_b, ws.e16, p.rows, H, d.mlp, EPI_DELTA16, VM_PROF_GEMM_RESID);
        }
        p.pend_proj = ws.d16;
        p.pend_fc2 = ws.e16;
        return r;
    };
    auto pool = [&](Pass &p) -> int {
        uint16_t *dst = (uint16_t *)out_emb + (size_t)p.b0 * e->out_dim;
        return vm_pool(ctx, dt, p.ws.x32, p.pend_proj, p.pend_fc2, e->ln_g, e->ln_b, d.ln_eps, e->proj_w, d.proj_dim,
                       l2_normalise, dst, p.nb, T, H, st);
    };
#define VM_TRY(x) do { if ((rc = (x)) != VM_OK) return rc; } while (0)
    auto run_passes = [&]() -> int {
        int pass = 0;
        for (int b0 = 0; b0 < B; ++pass) {
            Pass A;
            A.b0 = b0;
            A.nb = B - b0 < mb ? B - b0 : mb;
            A.rows = A.nb * T;
            A.ws = dual && (pass & 1) ? ws1 : ws0;
            st = dual ? e->side[pass & 1] : st0;    // the stage lambdas launch on `st`
            // developer experiment (VIDMEM_ENC_PHASE = n, measured null: DESIGN.md 4.6): the second stream starts behind
            // the n-th stage of the first pass's first layer instead of together with it
            static const int phase_env = (int)VM_DEV_ENV("ENC_PHASE", 0);
            const bool mark = dual && phase_env > 0 && pass == 0;
            if (dual && phase_env > 0 && pass == 1) (void)hipStreamWaitEvent(st, e->ev_phase, 0);
            VM_TRY(embed(A));
            for (int l = 0; l < d.layers; ++l) {
                VM_TRY(ln1(A, l));
                if (mark && l == 0 && phase_env == 1) (void)hipEventRecord(e->ev_phase, st);
                VM_TRY(attn_block(A, l));
                if (mark && l == 0 && phase_env == 2) (void)hipEventRecord(e->ev_phase, st);
                VM_TRY(ln2(A, l));
                if (mark && l == 0 && phase_env == 3) (void)hipEventRecord(e->ev_phase, st);
                VM_TRY(mlp_block(A, l));
                if (mark && l == 0 && phase_env == 4) (void)hipEventRecord(e->ev_phase, st);
                if (mark && l == 1 && phase_env == 5) (void)hipEventRecord(e->ev_phase, st);   // a layer and a half... (after layer 1's MLP)
            }
            VM_TRY(pool(A));
            b0 += A.nb;
        }
        return VM_OK;
    };
#undef VM_TRY
    rc = run_passes();
    if (dual) {
        // join, ALSO after a failed launch: kernels already queued on the side streams still use the workspaces, so the
        // caller's stream must not run past them (and a stream capture must not be left with unjoined forks)
        for (int i = 0; i < 2; ++i) {
            hipError_t he = hipEventRecord(e->ev_join[i], e->side[i]);
            if (he == hipSuccess) he = hipStreamWaitEvent(st0, e->ev_join[i], 0);
            if (he != hipSuccess && rc == VM_OK)
                rc = vm_fail(ctx, VM_ERR_HIP, "vm_encode: joining the internal streams failed: %s", hipGetErrorString(he));
        }
    }
    return rc;
}
