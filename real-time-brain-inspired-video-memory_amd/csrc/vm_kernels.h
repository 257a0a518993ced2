// Internal kernel launchers of the encoder (not part of the C ABI).
#pragma once
#include "vm_common.h"

enum {
    EPI_STORE16 = 0,  // out16[t, f] = acc + bias
    EPI_GELU16 = 1,   // out16 = gelu_erf(acc + bias)
    EPI_QGELU16 = 2,  // out16 = x * sigmoid(1.702 x)
    EPI_RESID32 = 3,  // out32[t, f] += acc + bias           (fp32 residual stream, in place)
    EPI_PATCH = 4,    // out32[frame*T + 1 + p, f] = acc + bias + pos[1 + p, f]   (patch embedding)
    EPI_DELTA16 = 5   // out16 = fp16(acc + bias) WHATEVER the operand dtype: a residual-branch output (patch rows,
                      // attention projection, FC2) is never a matrix operand - the LayerNorm / embed / pool kernels
                      // add it to the fp32 residual stream - so a bf16 encoder stores it with fp16's 11 bits
                      // (tests/golden/bf16_floor.py: 5.8e-3 -> 5.1e-3 on CLIP-L/14-336).  Same as STORE16 for fp16.
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
    int prof_cat;       // vm_prof_cat of this launch (bench.py's per-kernel breakdown)
    int head_major;     // 16-bit epilogues: out16 is [N/64][M][64] (per-head contiguous blocks) instead of [M, ldo]
    int hm_rows, hm_stride;  // head-major only: rows per head block (0 = M) and the block row of GEMM row t = t * hm_stride
                             // (0 = 1): lets a GEMM over the CLS rows alone write into the all-rows q blocks
    int stream_out;     // set by vm_gemm: the 16-bit output is larger than L2 and leaves with the non-temporal policy
    int fgroup;         // set by vm_gemm: feature tiles per group of the persistent kernel's tile order (0 = all)
    int explicit_zero;  // set by vm_gemm (developer A/B, VIDMEM_GEMM_ZERO=1): clear accumulators per tile instead of C = 0
    const float *gelu_tab;  // set by vm_gemm: the context's erf-GELU table (EPI_GELU16 only)
    unsigned long long *stamps;  // developer harness only (-DVM_GEMM_ABLATE builds read it): [grid][64][2] 100 MHz wall-clock
                                 // stamps of the persistent kernel's tile boundaries (K loop done / epilogue issued); else null
};

int vm_gemm(vm_ctx *ctx, int dtype, const GemmArgs &g, int epi, hipStream_t st);

// qkv HEAD-MAJOR 16-bit [3*heads][B*T][64] (block index = {q,k,v} * heads + head; GemmArgs::head_major) ->
// ctx [B*T, H] 16-bit row-major; head dim 64.
// q_rows > 0: only the first q_rows query rows of every frame are needed (rounded up to 16-row tiles; the other rows
// of ctx_out are left untouched)
int vm_attention(vm_ctx *ctx, int dtype, const uint16_t *qkv, uint16_t *ctx_out, int B, int T, int heads,
                 hipStream_t st, int q_rows = 0);

// v = (x32[row] + delta16[row]) + deltaB16[row] (fp16 whatever `dtype`: EPI_DELTA16; either may be null);
// x32[row] = v when write_x == 1, or when write_x = n > 1 and row % n == 0; out16[row] = LayerNorm(v) * gamma + beta
// rstride: row r of the pass lives at row r * rstride of every array (1: dense; tokens per frame: the CLS rows only)
int vm_resid_layernorm(vm_ctx *ctx, int dtype, float *x32, const uint16_t *delta16, const uint16_t *deltaB16,
                       int write_x, const float *gamma, const float *beta, float eps, uint16_t *out16, int rows, int H,
                       hipStream_t st, int rstride = 1, int lowreg = 0);
// x32[frame*T + tok] = (tok ? patch16[frame*(T-1) + tok-1] : cls) + pos[tok], then the optional pre-LayerNorm
int vm_embed(vm_ctx *ctx, int dtype, const uint16_t *patch16, const float *cls, const float *pos, const float *pre_g,
             const float *pre_b, float eps, int pre_ln, float *x32, int B, int T, int H, hipStream_t st);
// CLS row ((x32 + delta16) + deltaB16) -> final LayerNorm, optional projection, optional L2 normalisation, cast
int vm_pool(vm_ctx *ctx, int dtype, const float *x, const uint16_t *delta16, const uint16_t *deltaB16,
            const float *gamma, const float *beta, float eps, const uint16_t *proj_w, int proj_dim, int l2, uint16_t *out,
            int B, int T, int H, hipStream_t st);
