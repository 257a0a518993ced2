// Internal kernel launchers of the encoder (not part of the C ABI).
#pragma once
#include "vm_common.h"

enum {
    EPI_STORE16 = 0,  // out16[t, f] = acc + bias
    EPI_GELU16 = 1,   // out16 = gelu_erf(acc + bias)
    EPI_QGELU16 = 2,  // out16 = x * sigmoid(1.702 x)
    EPI_RESID32 = 3,  // out32[t, f] += acc + bias           (fp32 residual stream, in place)
    EPI_PATCH = 4     // out32[frame*T + 1 + p, f] = acc + bias + pos[1 + p, f]   (patch embedding)
};

struct GemmArgs {
    const uint16_t *X;  // [M, ldx] activations, K-contiguous
    const uint16_t *W;  // [N, K] weights (torch Linear layout)
    const float *bias;  // [N]
    uint16_t *out16;
    float *out32;
    const float *pos;   // EPI_PATCH only: [T, N]
    int M, N, K;
    int ldx, ldo;
    int P, T;           // EPI_PATCH only: patches per frame, tokens per frame
};

int vm_gemm(vm_ctx *ctx, int dtype, const GemmArgs &g, int epi, hipStream_t st);

// qkv [B*T, 3H] 16-bit (q | k | v, head h at columns h*64) -> ctx [B*T, H] 16-bit; head dim 64.
int vm_attention(vm_ctx *ctx, int dtype, const uint16_t *qkv, uint16_t *ctx_out, int B, int T, int heads,
                 hipStream_t st);

// x [rows, H] fp32 -> out [rows, H] 16-bit, LayerNorm(gamma, beta, eps)
int vm_layernorm16(vm_ctx *ctx, int dtype, const float *x, const float *gamma, const float *beta, float eps,
                   uint16_t *out, int rows, int H, hipStream_t st);
// in-place fp32 LayerNorm (CLIP pre-LN on the residual stream)
int vm_layernorm32_inplace(vm_ctx *ctx, float *x, const float *gamma, const float *beta, float eps, int rows, int H,
                           hipStream_t st);
// x[frame*T + 0, :] = cls + pos[0]
int vm_cls_rows(vm_ctx *ctx, float *x, const float *cls, const float *pos, int B, int T, int H, hipStream_t st);
// final LayerNorm of the CLS row, optional projection, optional L2 normalisation, cast
int vm_pool(vm_ctx *ctx, int dtype, const float *x, const float *gamma, const float *beta, float eps,
            const uint16_t *proj_w, int proj_dim, int l2, uint16_t *out, int B, int T, int H, hipStream_t st);
