// Dense layers of the vision encoder: out[t, f] = sum_k X[t, k] * W[f, k] (+ fused epilogue), 16-bit operands,
// fp32 MFMA accumulation.  MFMA-bound: 95.9 % (ViT-B/16) / 91.4 % (CLIP-L/14-336) of the encoder's FLOPs run here.
//
// In both kernels the WEIGHT tile is the MFMA A operand and the ACTIVATION tile the B operand, so a lane ends up
// with 4 consecutive features of one token: epilogue stores are 8-byte (16-bit out) or 16-byte (fp32 residual) row
// pieces with no LDS transpose.  Staging is global_load_lds_dwordx4 (16 B/lane, 1 KiB per wave instruction = 8 rows
// x 128 B) into a linear LDS image; the XOR swizzle chunk ^= (row & 7) is applied on the per-lane SOURCE address and
// again on the ds_read_b128 address (both sides or neither), which makes the fragment reads bank-conflict free.
// Workgroups are renumbered so that the feature tiles of one token panel land on the same XCD (its L2 then serves the
// panel's re-reads).
//
//   gemm256_kernel  256 (features) x 256 (tokens) x 64 tile, 8 waves (2 x 4), one workgroup per CU (128 KiB LDS).
//                   Each K-tile runs as 4 phases of {ds_read fragments | 16 MFMA}, separated by raw s_barriers; the
//                   two wave rows are offset by one barrier, so on every SIMD one wave issues MFMAs while its partner
//                   reads LDS / issues the next K-tile's LDS-DMA.  The next K-tile is staged into the other LDS buffer
//                   2 instructions per phase and retired region by region with counted vmcnt waits.
//   gemm128_kernel  128 x 128 x 64 tile, 4 waves, two workgroups per CU: used when a 256-row tile grid would leave
//                   most CUs idle (small batches).
#include "vm_internal.h"
#include "vm_kernels.h"
#include "gemm_guard.h"
#ifndef VM_GEMM_W_AUX
#define VM_GEMM_W_AUX 0
#endif
#ifndef VM_GEMM_X_AUX
#define VM_GEMM_X_AUX 0
#endif
#ifndef VM_GELU_POLY
#define VM_GELU_POLY 0   // 1: the round-2 erf-GELU (sigmoid of a polynomial, two transcendentals) for harness A/Bs
#endif

namespace {

constexpr int BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;  // 128 rows x 64 k, 16 KiB

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

// x * sigmoid(1.702 x) with one v_exp_f32 and one v_rcp_f32 (the IEEE division sequence cost ~10 more instructions per
// element; 1 ulp of the reciprocal is far below the 16-bit output's rounding)
__device__ __forceinline__ float quick_gelu(float x) {
    const float e = __builtin_amdgcn_exp2f(x * (-1.702f * 1.44269504088896340736f));
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// erf-GELU, x * Phi(x), as x * sigmoid(x * P(x^2)) with a degree-4 minimax P: logit(Phi(x)) is a smooth odd function, so
// five coefficients reach |error| <= 3.4e-6 ABSOLUTE over the whole line in fp32 (fit and check: DESIGN.md 4.2; the
// 16-bit output rounds at 5e-4 / 4e-3 relative), and the tails saturate by themselves (exp2 -> 0 / inf).  -log2(e) is
// folded into the coefficients: u = x*x, four fma, one mul, v_exp_f32, one add, v_rcp_f32, one mul = 12 VALU issue
// units per element against 18 for the previous form (erfc(z) = (1 + a1 z + ... + a6 z^6)^-16, A&S 7.1.28, with its
// sign select).  The FC1 epilogue evaluates 65,536 of these per tile with the matrix pipe idle: it is VALU-bound.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    const f32x2 u = x * x;
    f32x2 p = __builtin_elementwise_fma(u, f32x2{-3.229004050808726e-06f, -3.229004050808726e-06f},
                                        f32x2{8.823838288662955e-05f, 8.823838288662955e-05f});
    p = __builtin_elementwise_fma(p, u, f32x2{0.00036027334863319993f, 0.00036027334863319993f});
    p = __builtin_elementwise_fma(p, u, f32x2{-0.10522668808698654f, -0.10522668808698654f});
    p = __builtin_elementwise_fma(p, u, f32x2{-2.3020453453063965f, -2.3020453453063965f});
    const f32x2 t = x * p;  // = -log2(e) * logit(Phi(x))
    f32x2 r;
    r.x = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t.x));
    r.y = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t.y));
    return x * r;
}

// erf-GELU by TABLE (round 4).  The FC1 epilogue is VALU-bound: 65,536 GELUs per tile on two waves per SIMD with the
// matrix pipe idle, ~54 issue cycles each in the form above (two 8-cycle transcendentals, eight packed or plain
// 4-cycle instructions).  Phi(x) is smooth and bounded, so a piecewise-LINEAR table does it in six instructions and one
// LDS read per element - v_med3 (clamp to the table), v_fma (index in the mantissa of 2^23 + ...), v_mad_u32_u24
// (byte address), ds_read_b64 {a, b}, v_fma (Phi = a + b x), v_mul (x Phi) - with |error| <= 1.0e-6 |x| against the
// 3.4e-6 of the polynomial form (table: context.hip build_gelu_table, 1,281 entries of 1/128 over [-5, 5], 10 KiB of
// LDS per workgroup, copied from the context's device copy when the kernel starts).  x beyond the table takes the end
// entries (Phi = 0 / 1); a NaN indexes entry 0 and comes out as NaN x 0 = NaN.
__device__ __forceinline__ unsigned gelu_tab_addr(float x, unsigned tab_off) {
    const float xc = __builtin_amdgcn_fmed3f(x, -5.0f, 5.0f);
    const float m = __builtin_fmaf(xc, 128.0f, 8388608.0f + 640.0f);   // mantissa = round(128 x) + 640, ulp = 1
    unsigned addr;   // low 24 bits of m = the mantissa (the exponent's lsb is 0): x 8 + table base in one instruction
    asm("v_mad_u32_u24 %0, %1, 8, %2" : "=v"(addr) : "v"(m), "s"(tab_off));   // (hipcc turns __umul24(m, 8) + base into 3)
    return addr;
}
typedef float tab2 __attribute__((ext_vector_type(2)));   // {a, b}: a FLOAT pair (bit_cast of a vector ELEMENT reads element 0)
// y[e] = x[e] * fma(b[e], x[e], a[e]) for the four values of one accumulator block.  One asm statement: written in
// C++, hipcc packs neighbours into v_pk_fma_f32 / v_pk_mul_f32 behind three v_mov that line the {a, b} pairs up (28
// issue cycles per two elements instead of 16), and statement by statement it puts an s_nop behind each (same IEEE fma
// and product in every form).
__device__ __forceinline__ f32x4 gelu_tab_apply4(f32x4 x, const tab2 (&ab)[4]) {
    float y0, y1, y2, y3;
    asm("v_fma_f32 %0, %9, %4, %8\n\tv_fma_f32 %1, %11, %5, %10\n\tv_fma_f32 %2, %13, %6, %12\n\t"
        "v_fma_f32 %3, %15, %7, %14\n\t"
        "v_mul_f32 %0, %4, %0\n\tv_mul_f32 %1, %5, %1\n\tv_mul_f32 %2, %6, %2\n\tv_mul_f32 %3, %7, %3"
        : "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(ab[0].x), "v"(ab[0].y), "v"(ab[1].x), "v"(ab[1].y),
          "v"(ab[2].x), "v"(ab[2].y), "v"(ab[3].x), "v"(ab[3].y));
    return f32x4{y0, y1, y2, y3};
}
// four lookups issued and retired in ONE asm statement (nothing the compiler could slip between issue and wait)
__device__ __forceinline__ void gelu_tab_read4(tab2 (&ab)[4], const unsigned (&addr)[4]) {
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(ab[0]), "=&v"(ab[1]), "=&v"(ab[2]), "=&v"(ab[3])
                 : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3])
                 : "memory");
}
// pipelined form for the persistent kernel's epilogue: issue now, retire two blocks later (counted lgkmcnt; other LGKM
// traffic in flight only makes the wait stronger)
__device__ __forceinline__ void gelu_tab_issue4(tab2 (&ab)[4], const unsigned (&addr)[4]) {
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7"
                 : "=&v"(ab[0]), "=&v"(ab[1]), "=&v"(ab[2]), "=&v"(ab[3])
                 : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3])
                 : "memory");
}
template <int YOUNGER>
__device__ __forceinline__ void gelu_tab_retire4(tab2 (&ab)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(ab[0]), "+v"(ab[1]), "+v"(ab[2]), "+v"(ab[3]) : "n"(YOUNGER) : "memory");
}
// workgroup-wide copy of the table into LDS (before any LDS-DMA of the kernel is issued)
__device__ __forceinline__ void gelu_tab_to_lds(const float *tab, char *dst, int tid, int nthreads) {
    for (int i = tid; i < VM_GELU_TAB_BYTES / 16; i += nthreads)
        reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(tab)[i];
}

// EPI_DELTA16 stores a bf16 encoder's residual-branch outputs as fp16 (vm_kernels.h): values beyond fp16's range - the
// outlier activations a bf16 checkpoint may have been chosen for - saturate at +-65504 instead of becoming inf (which
// the LayerNorm behind it would turn into a NaN row).  A NaN stays a NaN (v_med3_f32 alone would return -65504 for
// it and hide a broken activation behind a finite, wrong embedding): v_med3_f32 + v_cmp_u_f32 + v_cndmask per element,
// bf16 encoders only, in epilogues that are not VALU-bound.
__device__ __forceinline__ float sat_f16_1(float v) {
    // one scalar constant, used twice (negated once): as __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f) one of the two
    // bounds lives in a VECTOR register for the whole kernel (a VOP3 reads one scalar), which made this instantiation
    // the only one past 224 VGPRs - the allocation step that leaves room for the other stream's LayerNorm (encoder.hip)
    float c;
    asm("v_med3_f32 %0, %1, %2, -%2" : "=v"(c) : "v"(v), "s"(65504.0f));
    return __builtin_isnan(v) ? v : c;
}
__device__ __forceinline__ f32x2 sat_f16(f32x2 v) { return f32x2{sat_f16_1(v.x), sat_f16_1(v.y)}; }

// Element index of out16[t, f]: row-major [M, ldo], or head-major [N/64][M][64] (each 64-feature head a contiguous
// [M, 64] block: what the attention kernel streams; see vm_kernels.h).
__device__ __forceinline__ size_t out16_index(const GemmArgs &g, int t, int f) {
    return g.head_major ? ((((size_t)(f >> 6) * g.hm_rows + (size_t)t * g.hm_stride) << 6) | (f & 63))
                        : (size_t)t * g.ldo + f;
}

// 16-bit outputs leave with the non-temporal (streaming) policy: a GEMM writes 130-530 MB that the next kernel reads
// long after the 32 MiB of L2 have turned over, and written with the default policy the 33 MB burst of one round of
// tiles evicts the weight and activation panels the other CUs are still re-reading (measured on the 86,877-token
// GEMMs, alternating order: QKV -12..14 %, FC1+GELU -4..7 %, projection -2..3 %, FC2 0..-2 %).
// Outputs that fit L2 (GemmArgs::stream_out == 0: a streaming session's few frames) keep the default policy, the next
// kernel finds them there.
typedef unsigned u32x4_nt __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_nt __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_out16(const GemmArgs &g, uint16_t *p, uint4 v) {
    if (g.stream_out) __builtin_nontemporal_store(u32x4_nt{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_nt *>(p));
    else *reinterpret_cast<uint4 *>(p) = v;
}
__device__ __forceinline__ void store_out16(const GemmArgs &g, uint16_t *p, uint2 v) {
    if (g.stream_out) __builtin_nontemporal_store(u32x2_nt{v.x, v.y}, reinterpret_cast<u32x2_nt *>(p));
    else *reinterpret_cast<uint2 *>(p) = v;
}

// XCD-aware renumbering (bijective for any grid size): ids that share (blockIdx % 8) become neighbours.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int xcd = orig & 7, qd = nwg >> 3, rm = nwg & 7;
    return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (orig >> 3);
}

// Epilogue of one token row t for NI feature tiles of a wave: the lane owns features fbase + 16*i + [0,4).
// All loads (bias, residual / position rows) are issued before the first use so they overlap instead of
// serialising on one s_waitcnt each.
template <int DT, int EPI, int NI>
__device__ __forceinline__ void epilogue_row(const GemmArgs &g, const f32x4 (&a)[NI], int t, int fbase,
                                             unsigned tab_off = 0) {
    using E = vm_elem<(EPI == EPI_DELTA16) ? VM_F16 : DT>;  // output element type
    float4 b4[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) b4[i] = *reinterpret_cast<const float4 *>(g.bias + fbase + 16 * i);
    if (EPI == EPI_STORE16 || EPI == EPI_DELTA16 || EPI == EPI_GELU16 || EPI == EPI_QGELU16) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            // the same packed sequence as the 256x256 kernel's epilogue (bit-identical outputs across kernels)
            f32x2 v01 = f32x2{a[i][0], a[i][1]} + f32x2{b4[i].x, b4[i].y};
            f32x2 v23 = f32x2{a[i][2], a[i][3]} + f32x2{b4[i].z, b4[i].w};
            if (EPI == EPI_GELU16) {
#if VM_GELU_POLY
                v01 = gelu_erf2(v01);
                v23 = gelu_erf2(v23);
#else
                const unsigned ad[4] = {gelu_tab_addr(v01.x, tab_off), gelu_tab_addr(v01.y, tab_off),
                                        gelu_tab_addr(v23.x, tab_off), gelu_tab_addr(v23.y, tab_off)};
                tab2 ab[4];
                gelu_tab_read4(ab, ad);
                const f32x4 y = gelu_tab_apply4(f32x4{v01.x, v01.y, v23.x, v23.y}, ab);
                v01 = f32x2{y[0], y[1]};
                v23 = f32x2{y[2], y[3]};
#endif
            }
            if (EPI == EPI_QGELU16) {
                v01 = f32x2{quick_gelu(v01.x), quick_gelu(v01.y)};
                v23 = f32x2{quick_gelu(v23.x), quick_gelu(v23.y)};
            }
            if (EPI == EPI_DELTA16) {
                v01 = sat_f16(v01);
                v23 = sat_f16(v23);
            }
            store_out16(g, g.out16 + out16_index(g, t, fbase + 16 * i),
                        make_uint2(E::pack2(v01.x, v01.y), E::pack2(v23.x, v23.y)));
        }
    } else if (EPI == EPI_RESID32) {
        float *orow = g.out32 + (size_t)t * g.ldo + fbase;
        float4 r[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) r[i] = *reinterpret_cast<const float4 *>(orow + 16 * i);
#pragma unroll
        for (int i = 0; i < NI; ++i)
            *reinterpret_cast<float4 *>(orow + 16 * i) =
                make_float4(r[i].x + (a[i][0] + b4[i].x), r[i].y + (a[i][1] + b4[i].y), r[i].z + (a[i][2] + b4[i].z),
                            r[i].w + (a[i][3] + b4[i].w));
    } else {  // EPI_PATCH: GEMM row = frame*P + p  ->  token row frame*T + 1 + p, plus pos[1 + p]
        const int fr = t / g.P, p = t - fr * g.P;
        const float *prow = g.pos + (size_t)(1 + p) * g.N + fbase;
        float *orow = g.out32 + ((size_t)fr * g.T + 1 + p) * g.ldo + fbase;
        float4 r[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) r[i] = *reinterpret_cast<const float4 *>(prow + 16 * i);
#pragma unroll
        for (int i = 0; i < NI; ++i)
            *reinterpret_cast<float4 *>(orow + 16 * i) =
                make_float4((a[i][0] + b4[i].x) + r[i].x, (a[i][1] + b4[i].y) + r[i].y, (a[i][2] + b4[i].z) + r[i].z,
                            (a[i][3] + b4[i].w) + r[i].w);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 16-bit epilogue of the persistent kernel: bias (+ activation), pack, wave-private LDS transpose, stores of
// 4 token rows x 256 contiguous bytes.  acc[i][p][e] = out[token tbase + 16 p + r16][feature fw + 16 i + 4 h + e].
// Why through LDS (round 4, tools/store_probe.hip): a CU's store path takes a 16-byte-per-lane store at full rate only
// when CONSECUTIVE LANES write consecutive addresses (4 rows x 256 B per instruction: 128 KiB in 2,350 ticks alone on
// the chip); 16 rows x 64 B and 8 rows x 128 B - what lane-row exchanges (v_permlane16_swap + DPP row_ror:8) can build
// from the MFMA layout without LDS, bit-identical outputs - take 9,050 ticks, and the GEMM with them was 0-8 % slower
// (tools/experiments/gemm_epilogue_lane_swap_and_store_policy.patch).  Cache policy of these stores: nt = sc1 nt <
// plain = sc0 < sc1 = sc0 sc1 (same patch; QKV 682 / 684 / 724 / 725 / 722 / 723 us).
//
// Its LDS traffic (bias table, transpose scratch) is issued by INLINE ASM.  Written as plain C++ accesses, hipcc
// puts an s_waitcnt vmcnt(0) in front of the first of them: with LDS-DMA in flight it cannot prove that the access
// does not alias a DMA destination, so every tile's epilogue began by draining the staging of the NEXT tile's first
// K-tiles - a full memory latency with the matrix pipe idle, once per tile.  (The fragment reads of the K loops do
// not get that wait)  The asm reads carry no dependency the compiler can see, hence the explicit
// s_waitcnt lgkmcnt with the destination registers as "+v" operands before each first use.
// ---------------------------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ void lds_rd128(u32x4 &d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_wr64(unsigned addr, u32x2 v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned lds_offset(const void *p) {  // byte offset of a __shared__ address inside LDS
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char *)p;
}

// bias_w: the wave's 128 bias values in LDS (the tile's slot + wave row x 128); tab_off: LDS byte offset of the GELU table
template <int DT, int EPI, int NP, bool NOSTORE = false>
__device__ __forceinline__ void epilogue16(const GemmArgs &g, const f32x4 (&acc)[8][NP], const float *bias_w,
                                           const char *scratch, int tbase, int fw, int lane, unsigned tab_off) {
    using EO = vm_elem<(EPI == EPI_DELTA16) ? VM_F16 : DT>;  // output element type
    const int r16 = lane & 15, h = lane >> 4;
    const unsigned b_off = lds_offset(bias_w + 4 * h);
    const unsigned s_off = lds_offset(scratch);
    u32x4 braw[8];
    lds_rd128<0>(braw[0], b_off);
    lds_rd128<64>(braw[1], b_off);
    lds_rd128<128>(braw[2], b_off);
    lds_rd128<192>(braw[3], b_off);
    lds_rd128<256>(braw[4], b_off);
    lds_rd128<320>(braw[5], b_off);
    lds_rd128<384>(braw[6], b_off);
    lds_rd128<448>(braw[7], b_off);
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(braw[0]), "+v"(braw[1]), "+v"(braw[2]), "+v"(braw[3]), "+v"(braw[4]), "+v"(braw[5]),
                   "+v"(braw[6]), "+v"(braw[7]));
    const int wrow = r16 & 7, wsw = wrow << 4;
    const unsigned w_off = s_off + wrow * 256;
    const int row0 = lane >> 4, ch = lane & 15;   // transposed read: rows row0 and row0 + 4 of the 8-row pass
    const unsigned r_off0 = s_off + row0 * 256 + ((ch ^ row0) << 4);
    const unsigned r_off1 = s_off + (row0 + 4) * 256 + ((ch ^ (row0 + 4)) << 4);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        u32x2 pk[8];
        if (EPI == EPI_GELU16 && !VM_GELU_POLY) {
            // table GELU, software-pipelined over the 8 feature blocks: block i's four lookups are issued two blocks
            // before they are used, so their LDS latency (and bank conflicts: the addresses are data) hides under the
            // index arithmetic and the products of the neighbouring blocks
            f32x4 v[3];
            tab2 ab[3][4];
            auto issue = [&](int i) {
                const f32x4 a = acc[i][p];
                const f32x4 b = __builtin_bit_cast(f32x4, braw[i]);
                const f32x2 v01 = f32x2{a[0], a[1]} + f32x2{b[0], b[1]};  // v_pk_add_f32
                const f32x2 v23 = f32x2{a[2], a[3]} + f32x2{b[2], b[3]};
                v[i % 3] = f32x4{v01.x, v01.y, v23.x, v23.y};
                const unsigned ad[4] = {gelu_tab_addr(v01.x, tab_off), gelu_tab_addr(v01.y, tab_off),
                                        gelu_tab_addr(v23.x, tab_off), gelu_tab_addr(v23.y, tab_off)};
                gelu_tab_issue4(ab[i % 3], ad);
            };
            issue(0);
            issue(1);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i + 2 < 8) issue(i + 2);
                if (i + 2 < 8) gelu_tab_retire4<8>(ab[i % 3]);
                else if (i + 1 < 8) gelu_tab_retire4<4>(ab[i % 3]);
                else gelu_tab_retire4<0>(ab[i % 3]);
                const f32x4 y = gelu_tab_apply4(v[i % 3], ab[i % 3]);
                pk[i] = u32x2{EO::pack2(y[0], y[1]), EO::pack2(y[2], y[3])};
            }
        } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 a = acc[i][p];
            const f32x4 b = __builtin_bit_cast(f32x4, braw[i]);
            f32x2 v01 = f32x2{a[0], a[1]} + f32x2{b[0], b[1]};  // v_pk_add_f32
            f32x2 v23 = f32x2{a[2], a[3]} + f32x2{b[2], b[3]};
            if (EPI == EPI_GELU16) {
                v01 = gelu_erf2(v01);
                v23 = gelu_erf2(v23);
            }
            if (EPI == EPI_QGELU16) {
                v01 = f32x2{quick_gelu(v01.x), quick_gelu(v01.y)};
                v23 = f32x2{quick_gelu(v23.x), quick_gelu(v23.y)};
            }
            if (EPI == EPI_DELTA16) {
                v01 = sat_f16(v01);
                v23 = sat_f16(v23);
            }
            pk[i] = u32x2{EO::pack2(v01.x, v01.y), EO::pack2(v23.x, v23.y)};  // v_cvt_pk_*
        }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if ((r16 >> 3) == half) {
#pragma unroll
                for (int i = 0; i < 8; ++i) lds_wr64(w_off + ((32 * i + 8 * h) ^ wsw), pk[i]);
            }
            u32x4 v0, v1;
            lds_rd128<0>(v0, r_off0);
            lds_rd128<0>(v1, r_off1);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1));
            const int t = tbase + 16 * p + 8 * half + row0;
            if (NOSTORE) {
                asm volatile("" ::"v"(v0), "v"(v1));
            } else {
                if (t < g.M) store_out16(g, g.out16 + out16_index(g, t, fw + ch * 8), make_uint4(v0.x, v0.y, v0.z, v0.w));
                if (t + 4 < g.M)
                    store_out16(g, g.out16 + out16_index(g, t + 4, fw + ch * 8), make_uint4(v1.x, v1.y, v1.z, v1.w));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 256 x 256 x 64, 8 waves, 4 phases per K-tile
// ---------------------------------------------------------------------------------------------------------------
template <int DT, int EPI, int ABL = 0>  // ABL: developer ablation bits (1 no LDS-DMA in loop, 2 no ds_read, 4 no MFMA)
__global__ void __launch_bounds__(512, 1) gemm256_kernel(GemmArgs g) {
    using E = vm_elem<DT>;
    using vec8 = typename E::vec8;
    // [2 buffers][W half0 | W half1 | X half0 | X half1], each half 128 rows x 128 B
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, h = lane >> 4;
    const int tiles_n = g.N >> 8;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int t0 = tm << 8, f0 = tn << 8;
    const int wr = wave >> 2, wc = wave & 3;  // wave row: 128 features; wave column: 64 tokens
    const int K = g.K, M = g.M;
    constexpr bool TAB = EPI == EPI_GELU16 && !VM_GELU_POLY;
    if (TAB) gelu_tab_to_lds(g.gelu_tab, smem + 8 * HALF_BYTES, tid, 512);   // visible after the K loop's barriers
    const unsigned tab_off = lds_offset(smem + 8 * HALF_BYTES);

    // staging: per K-tile every wave issues 8 LDS-DMA instructions of 8 rows x 128 B, two from each REGION, where a
    // region is the set of rows all waves read in the same phase:
    //   Wa0 = W rows {0..63, 128..191} (phase 1)   Wa1 = W rows {64..127, 192..255} (phase 3)
    //   Xb0 = X rows {64c + 0..31}     (phase 1)   Xb1 = X rows {64c + 32..63}      (phase 2)      c = 0..3
    // 8-row block q = 2*wave + u (u = 0, 1) of a region; LDS byte offset of tile row r is r * 128.
    const int srow = lane >> 3, scp = lane & 7;
    const uint16_t *src_wa0[2], *src_wa1[2], *src_xb0[2], *src_xb1[2];
    int lds_wa0[2], lds_xb0[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int q = 2 * wave + u;
        const int wrow0 = (q < 8 ? q * 8 : 128 + (q - 8) * 8);  // first row of the block in Wa0; Wa1 = +64
        const int xrow0 = (q >> 2) * 64 + (q & 3) * 8;          // first row of the block in Xb0; Xb1 = +32
        const int wrow = wrow0 + srow, xrow = xrow0 + srow;     // (row & 7) == srow in every region
        const int chunk = scp ^ srow;
        src_wa0[u] = g.W + (size_t)(f0 + wrow) * K + chunk * 8;
        src_wa1[u] = g.W + (size_t)(f0 + wrow + 64) * K + chunk * 8;
        int t_b0 = t0 + xrow, t_b1 = t0 + xrow + 32;
        if (t_b0 > M - 1) t_b0 = M - 1;
        if (t_b1 > M - 1) t_b1 = M - 1;
        src_xb0[u] = g.X + (size_t)t_b0 * g.ldx + chunk * 8;
        src_xb1[u] = g.X + (size_t)t_b1 * g.ldx + chunk * 8;
        lds_wa0[u] = wrow0 * 128;
        lds_xb0[u] = 2 * HALF_BYTES + xrow0 * 128;
    }
    auto dma2 = [&](const uint16_t *const (&src)[2], const int (&dst)[2], int extra, int kt, int buf) {
        if ((ABL & 1) && kt > 0) return;
#pragma unroll
        for (int u = 0; u < 2; ++u)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[u] + kt * BK),
                                             (lds_ptr_t)(smem + buf * 4 * HALF_BYTES + dst[u] + extra), 16, 0, 0);
    };
    auto stage_Wa0 = [&](int kt, int buf) { dma2(src_wa0, lds_wa0, 0, kt, buf); };
    auto stage_Wa1 = [&](int kt, int buf) { dma2(src_wa1, lds_wa0, 64 * 128, kt, buf); };
    auto stage_Xb0 = [&](int kt, int buf) { dma2(src_xb0, lds_xb0, 0, kt, buf); };
    auto stage_Xb1 = [&](int kt, int buf) { dma2(src_xb1, lds_xb0, 32 * 128, kt, buf); };

    // fragment read addresses inside one buffer (byte offsets); row & 7 == r16 & 7 for every 16-row tile
    const int sw0 = ((h ^ (r16 & 7)) << 4), sw1 = (((h + 4) ^ (r16 & 7)) << 4);
    const int a_base = (wr * 128 + r16) * 128;                       // + i*16*128, i = 0..7
    const int b_base = 2 * HALF_BYTES + (wc * 64 + r16) * 128;       // + j*16*128, j = 0..3

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    vec8 af[4][2], bf[4][2];  // af: current half (4 feature tiles) x 2 k-substeps; bf: all 4 token tiles

    auto read_a = [&](const char *buf, int half) {
        if ((ABL & 2) && buf != smem) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char *p = buf + a_base + (half * 4 + i) * 2048;
            af[i][0] = *reinterpret_cast<const vec8 *>(p + sw0);
            af[i][1] = *reinterpret_cast<const vec8 *>(p + sw1);
        }
    };
    auto read_b = [&](const char *buf, int half) {
        if ((ABL & 2) && buf != smem) return;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const char *p = buf + b_base + (half * 2 + j) * 2048;
            bf[half * 2 + j][0] = *reinterpret_cast<const vec8 *>(p + sw0);
            bf[half * 2 + j][1] = *reinterpret_cast<const vec8 *>(p + sw1);
        }
    };
    auto mma = [&](int ahalf, int bhalf) {
        if (ABL & 4) return;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ahalf * 4 + i][bhalf * 2 + j] =
                        E::mfma16(af[i][s], bf[bhalf * 2 + j][s], acc[ahalf * 4 + i][bhalf * 2 + j]);
        __builtin_amdgcn_s_setprio(0);
    };
#define VM_BAR() __builtin_amdgcn_s_barrier()
#define VM_LGKM0()                                         \
    do {                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                 \
    } while (0)

    // LDS-DMA of K-tile kt+1 is spread over the four phases of K-tile kt, 2 instructions each, issue order
    // Wa0, Xb0, Xb1, Wa1.  Each region is retired by a COUNTED wait placed before the first barrier of the phase
    // that precedes its first read (never vmcnt(0) in the steady state):
    //   phase 4 of kt  : vmcnt(4) leaves {Xb1, Wa1} in flight, retires Wa0 + Xb0 -> read in phase 1 of kt+1
    //   phase 1 of kt+1: vmcnt(4) after issuing the next Wa0: retires Xb1          -> read in phase 2
    //   phase 2 of kt+1: vmcnt(4) after issuing the next Xb0: retires Wa1          -> read in phase 3
    const int nk = K / BK;
    stage_Wa0(0, 0);
    stage_Xb0(0, 0);
    stage_Xb1(0, 0);
    stage_Wa1(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VM_BAR();
    if (wr == 1) VM_BAR();  // wave row 1 runs one barrier behind wave row 0

    for (int kt = 0; kt < nk; ++kt) {
        const char *buf = smem + (kt & 1) * 4 * HALF_BYTES;
        const bool more = kt + 1 < nk;
        const int nb = (kt + 1) & 1;
        // phase 1
        read_a(buf, 0);
        read_b(buf, 0);
        if (more) {
            stage_Wa0(kt + 1, nb);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        VM_BAR();
        VM_LGKM0();
        mma(0, 0);
        VM_BAR();
        // phase 2
        read_b(buf, 1);
        if (more) {
            stage_Xb0(kt + 1, nb);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        VM_BAR();
        VM_LGKM0();
        mma(0, 1);
        VM_BAR();
        // phase 3
        read_a(buf, 1);
        if (more) stage_Xb1(kt + 1, nb);
        VM_BAR();
        VM_LGKM0();
        mma(1, 1);
        VM_BAR();
        // phase 4
        if (more) {
            stage_Wa1(kt + 1, nb);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        VM_BAR();
        mma(1, 0);
        VM_BAR();
    }
    if (wr == 0) VM_BAR();  // match wave row 1's extra barrier
#undef VM_BAR
#undef VM_LGKM0

    // epilogue: acc[i][j][e] = out[token t0 + wc*64 + 16j + r16][feature f0 + wr*128 + 16i + 4h + e]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = t0 + wc * 64 + j * 16 + r16;
        if (t >= M) continue;
        f32x4 col[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) col[i] = acc[i][j];
        if (ABL & 8) {  // developer ablation: keep the accumulators live, skip the stores
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(col[i]));
            continue;
        }
        epilogue_row<DT, EPI, 8>(g, col, t, f0 + wr * 128 + 4 * h, tab_off);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent form of gemm256: one workgroup per CU walks tiles bid, bid+grid, ...  The K-tile stream continues across
// tile boundaries: the first K-tile of the NEXT output tile is staged during the last K-tile of the current one, and
// the epilogue's stores are left in flight (counted vmcnt) while the next tile's MFMAs start, so neither the
// prologue latency nor the output write-back leaves the matrix pipe idle between tiles.
// ---------------------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int DT, int EPI, int ABL = 0>  // ABL (developer ablation): 8 = no global stores, 16 = no epilogue at all
__global__ void __launch_bounds__(512, 1) gemm256p_kernel(GemmArgs g) {
    using E = vm_elem<DT>;
    using vec8 = typename E::vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // vector-memory operations an epilogue leaves behind the last LDS-DMA (its stores; loads are consumed before)
    constexpr bool OUT16 = EPI == EPI_STORE16 || EPI == EPI_DELTA16 || EPI == EPI_GELU16 || EPI == EPI_QGELU16;
    constexpr int EPI_STORES = OUT16 ? 16 : 32;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, h = lane >> 4;
    const int tiles_n = g.N >> 8;
    const int panels = (g.M + 255) >> 8;
    const int ntiles = panels * tiles_n;
    const int vbid = xcd_remap(blockIdx.x, gridDim.x);
    const int wr = wave >> 2, wc = wave & 3;
    const int K = g.K, M = g.M;
    const int nk = K / BK;
    // Tile order.  Tiles are dealt to the workgroups in rounds of gridDim.x, 32 consecutive ones to an XCD.  With the
    // feature tiles of a token panel consecutive (fgroup == tiles_n) an XCD touches EVERY weight tile in every round:
    // weights larger than its 4 MiB L2 (FC1: 4.7 MB, QKV: 3.5 MB beside the activation stream) are re-fetched from the
    // Infinity Cache once per round (PMC, round 2: FC1 604 MB of 1,316 MB fabric traffic per launch).  So the feature
    // tiles go in GROUPS of `fgroup` whose weights fit an L2: all token panels against group 0, then all against group
    // 1, ...: the weights of a group come in once per XCD, the activations are streamed once per group instead.
    const int fgroup = g.fgroup > 0 && g.fgroup < tiles_n ? g.fgroup : tiles_n;
    auto tile_coords = [&](int tile, int &tm, int &tn) {
        const int per_group = panels * fgroup;           // tiles of a full group (only the last one may be smaller)
        const int grp = tile / per_group;
        const int r = tile - grp * per_group;
        int gsz = tiles_n - grp * fgroup;
        if (gsz > fgroup) gsz = fgroup;
        tm = r / gsz;
        tn = grp * fgroup + (r - tm * gsz);
    };

    // Staging by BUFFER loads to LDS (buffer_load_dwordx4 ... offen lds): the per-lane part of a source address is a
    // 32-bit byte offset that never changes (row of the lane inside the tile, swizzled chunk), the tile origin sits in
    // the descriptor's base and the K-tile in the wave-uniform soffset.  Against global_load_lds with 64-bit per-lane
    // pointers this drops eight 64-bit pointers (16 VGPRs) and the per-tile pointer arithmetic (time: equal within
    // 1 %, the projection GEMM 5 % faster).  Token rows past M need no clamp: their per-lane offset lies past the
    // tile descriptor's num_records, the range check returns zeros, and their outputs are never stored.
    const int srow = lane >> 3, scp = lane & 7;
    const int chunk = scp ^ srow;
    unsigned voff_w[2], voff_x[2];   // byte offsets of this lane's 16 bytes inside the W / X tile, region *a0* / *b0*
    int lds_wa0[2], lds_xb0[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int q = 2 * wave + u;
        const int wrow0 = (q < 8 ? q * 8 : 128 + (q - 8) * 8);
        const int xrow0 = (q >> 2) * 64 + (q & 3) * 8;
        voff_w[u] = (unsigned)(((wrow0 + srow) * K + chunk * 8) * 2);
        voff_x[u] = (unsigned)(((xrow0 + srow) * g.ldx + chunk * 8) * 2);
        lds_wa0[u] = wrow0 * 128;
        lds_xb0[u] = 2 * HALF_BYTES + xrow0 * 128;
    }
    // One descriptor per operand and TILE: base = the tile's first row, num_records = the bytes of the tile's rows that
    // exist (gemm_guard.h bounds them below 2^31), so (a) nothing here is a 32-bit offset from the start of a tensor -
    // any M, any row stride - and (b) a row past M is out of range by its PER-LANE offset alone: the wave-uniform
    // soffset only ever carries the K-tile (128 B x kt, inside the row), whatever the hardware's range check makes of
    // an soffset.
    __amdgpu_buffer_rsrc_t rs_w, rs_x;
    const unsigned w64 = (unsigned)(64 * K * 2), x32 = (unsigned)(32 * g.ldx * 2);   // region *1 = region *0 + 64 / 32 rows
    auto set_sources = [&](int tile) {
        int tm, tn;
        tile_coords(tile, tm, tn);
        const int f0s = (ABL & 64) ? 0 : (tn << 8);     // ablation: every tile reads W tile 0
        const int t0s = (ABL & 32) ? 0 : (tm << 8);     // ablation: ... token panel 0
        int vt = M - t0s;
        if (vt > 256) vt = 256;
        rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)(g.W + (size_t)f0s * K), 0, 256 * K * 2, 0x00020000);
        // the last existing row ends K elements in, not ldx (a strided view's allocation may stop there)
        rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)(g.X + (size_t)t0s * g.ldx), 0,
                                                 (int)(((size_t)(vt - 1) * g.ldx + K) * 2), 0x00020000);
    };
    // cache policy of the operand loads (aux bits of the buffer load: 2 = nt) is a build-time switch for the probe
#define VM_DMA2(NAME, AUX)                                                                                            \
    auto NAME = [&](const __amdgpu_buffer_rsrc_t rs, const unsigned (&voff)[2], unsigned vextra, unsigned soff,       \
                    const int (&dst)[2], int extra, int buf) {                                                        \
        _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + buf * 4 * HALF_BYTES + dst[u] + extra),   \
                                                     16, voff[u] + vextra, soff, 0, AUX);                             \
    }
    VM_DMA2(dma2w, VM_GEMM_W_AUX);
    VM_DMA2(dma2x, VM_GEMM_X_AUX);
#undef VM_DMA2
    auto stage_Wa0 = [&](int kt, int buf) { dma2w(rs_w, voff_w, 0u, kt * (BK * 2), lds_wa0, 0, buf); };
    auto stage_Wa1 = [&](int kt, int buf) { dma2w(rs_w, voff_w, w64, kt * (BK * 2), lds_wa0, 64 * 128, buf); };
    auto stage_Xb0 = [&](int kt, int buf) { dma2x(rs_x, voff_x, 0u, kt * (BK * 2), lds_xb0, 0, buf); };
    auto stage_Xb1 = [&](int kt, int buf) { dma2x(rs_x, voff_x, x32, kt * (BK * 2), lds_xb0, 32 * 128, buf); };

    const int sw0 = ((h ^ (r16 & 7)) << 4), sw1 = (((h + 4) ^ (r16 & 7)) << 4);
    const int a_base = (wr * 128 + r16) * 128;
    const int b_base = 2 * HALF_BYTES + (wc * 64 + r16) * 128;

    f32x4 acc[8][4];
    vec8 af[4][2], bf[4][2];
    auto read_a = [&](const char *buf, int half) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char *p = buf + a_base + (half * 4 + i) * 2048;
            af[i][0] = *reinterpret_cast<const vec8 *>(p + sw0);
            af[i][1] = *reinterpret_cast<const vec8 *>(p + sw1);
        }
    };
    auto read_b = [&](const char *buf, int half) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const char *p = buf + b_base + (half * 2 + j) * 2048;
            bf[half * 2 + j][0] = *reinterpret_cast<const vec8 *>(p + sw0);
            bf[half * 2 + j][1] = *reinterpret_cast<const vec8 *>(p + sw1);
        }
    };
    // first: the first K-step of a tile starts from C = 0 inside the MFMA (inline constant) instead of 128 cleared
    // registers per wave and tile (zero_acc: ~2 % of a tile's time in vector moves)
    const bool c0_start = g.explicit_zero == 0;   // developer A/B (VIDMEM_GEMM_ZERO=1): clear the accumulators instead
    auto clear_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto mma = [&](int ahalf, int bhalf, bool first) {
        __builtin_amdgcn_s_setprio(1);
        if (first && c0_start) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ahalf * 4 + i][bhalf * 2 + j] =
                        E::mfma16(af[i][0], bf[bhalf * 2 + j][0], f32x4{0.f, 0.f, 0.f, 0.f});
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ahalf * 4 + i][bhalf * 2 + j] =
                        E::mfma16(af[i][0], bf[bhalf * 2 + j][0], acc[ahalf * 4 + i][bhalf * 2 + j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[ahalf * 4 + i][bhalf * 2 + j] =
                    E::mfma16(af[i][1], bf[bhalf * 2 + j][1], acc[ahalf * 4 + i][bhalf * 2 + j]);
        __builtin_amdgcn_s_setprio(0);
    };
#define VM_BAR() __builtin_amdgcn_s_barrier()
#define VM_LGKM0()                                         \
    do {                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                 \
    } while (0)

    int tile = vbid;
    if (tile >= ntiles) return;  // whole workgroup leaves together (grid <= ntiles, so this never splits a barrier)
    // (Round 4 measured two start-delay switches here - per XCD slot and a ramp inside every XCD - null on every shape:
    // the CUs' tile boundaries spread over more than a tile period by themselves after a few rounds, DESIGN.md 4.2.  The
    // code is gone: its three live registers took the kernel from 224 to 227 VGPRs, past the allocation step that leaves
    // 64 registers per SIMD for the low-register LayerNorm of the encoder's other stream, encoder.hip.)
    // LDS behind the staging buffers: [2 bias slots of 1 KiB][8 x 2 KiB epilogue scratch][GELU table].
    // The 256 bias values of a TILE arrive by one LDS-DMA instruction (64 lanes x 16 B, wave 0) in the slot of the
    // tile's parity, issued a whole tile ahead (prologue: tile 0; boundary n: tile n + 1, ahead of the prestage), so the
    // epilogue needs no vector-memory load (it would queue behind the LDS-DMA in flight and expose its full latency
    // once per tile) and the whole bias vector no longer takes the 12-16 KiB the table now needs.  One more operation
    // in wave 0's in-order stream only makes its counted waits stronger.
    char *bias_slots = smem + 8 * HALF_BYTES;
    char *scratch_base = bias_slots + 2048;
    constexpr bool TAB = EPI == EPI_GELU16 && !VM_GELU_POLY;
    if (TAB) gelu_tab_to_lds(g.gelu_tab, scratch_base + 16384, tid, 512);   // before any LDS-DMA: plain stores
    const unsigned tab_off = lds_offset(scratch_base + 16384);
    auto stage_bias = [&](int tile_, int slot) {
        if (OUT16 && wave == 0) {
            int tm_, tn_;
            tile_coords(tile_, tm_, tn_);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g.bias + (tn_ << 8) + lane * 4),
                                             (lds_ptr_t)(bias_slots + slot * 1024), 16, 0, 0);
        }
    };
    int tcount = 0;   // tiles finished by this workgroup (bias slot = parity)
    stage_bias(tile, 0);
    set_sources(tile);
    stage_Wa0(0, 0);
    stage_Xb0(0, 0);
    stage_Xb1(0, 0);
    stage_Wa1(0, 0);
    wait_vmcnt<0>();
    VM_BAR();
    if (wr == 1) VM_BAR();  // wave row 1 runs one barrier behind wave row 0
    if (!c0_start) clear_acc();

    int gk = 0;  // K-tiles consumed so far (selects the LDS buffer)
    // State carried across a tile boundary.  The vector-memory pipe of a CU is in order, so the epilogue's 128 KiB of
    // stores must not sit in front of the next K-tiles' LDS-DMA: at the boundary the SECOND K-tile of the next tile
    // is staged BEFORE the epilogue (the first was staged during the last K-tile), and the counted waits below skip
    // over the stores, which then drain under two K-tiles of MFMA work.
    //   ep = 0: no epilogue behind        ep = 1: epilogue of a full tile (exactly EPI_STORES stores per wave)
    //   ep = 2: epilogue with an unknown (smaller) store count (ragged last row panel, fp32 epilogues): waits assume
    //           zero stores, i.e. they also wait for the stores - always safe, rare
    int ep = 0;
    bool prestaged = false;  // K-tile 1 of the current tile was staged at the boundary
#define VM_WAIT_EP(N)                                        \
    do {                                                     \
        if (ep == 1) wait_vmcnt<(N) + EPI_STORES>();         \
        else wait_vmcnt<(N)>();                              \
    } while (0)
    while (true) {
        const int next_tile = tile + gridDim.x;
        const bool has_next = next_tile < ntiles;
        for (int kt = 0; kt < nk; ++kt, ++gk) {
            const char *buf = smem + (gk & 1) * 4 * HALF_BYTES;
            const int nb = (gk + 1) & 1;
            int skt = kt + 1;
            bool more = skt < nk;
            if (!more && has_next) {  // roll over to the next output tile's first K-tile
                set_sources(next_tile);
                skt = 0;
                more = true;
            }
            const bool k0_after = prestaged && kt == 0;  // next K-tile already in flight: nothing to stage here
            const bool k1_after = prestaged && kt == 1;  // stages K-tile 2 as usual, counts skip the stores
            // phase 1
            read_a(buf, 0);
            read_b(buf, 0);
            if (k0_after) {
                VM_WAIT_EP(10);  // retire Xb1 of this K-tile; younger: its Wa1 (2) + K-tile 1 (8) [+ stores]
            } else if (more) {
                stage_Wa0(skt, nb);
                if (k1_after) VM_WAIT_EP(4); else wait_vmcnt<4>();
            } else {
                wait_vmcnt<0>();
            }
            VM_BAR();
            VM_LGKM0();
            mma(0, 0, kt == 0);
            VM_BAR();
            // phase 2
            read_b(buf, 1);
            if (k0_after) {
                VM_WAIT_EP(8);   // retire Wa1 of this K-tile; younger: K-tile 1 (8) [+ stores]
            } else if (more) {
                stage_Xb0(skt, nb);
                if (k1_after) VM_WAIT_EP(4); else wait_vmcnt<4>();
            }
            VM_BAR();
            VM_LGKM0();
            mma(0, 1, kt == 0);
            VM_BAR();
            // phase 3
            read_a(buf, 1);
            if (more && !k0_after) stage_Xb1(skt, nb);
            VM_BAR();
            VM_LGKM0();
            mma(1, 1, kt == 0);
            VM_BAR();
            // phase 4
            if (k0_after) {
                VM_WAIT_EP(4);   // retire Wa0 + Xb0 of K-tile 1; younger: its Xb1, Wa1 (4) [+ stores]
            } else if (more) {
                stage_Wa1(skt, nb);
                // from K-tile 1 on this also retires the epilogue's stores (>= 2 K-tiles old).  They cannot be given
                // longer: the count is in issue order, so waiting for K-tile 2's data - staged BEHIND the stores - waits
                // for the stores too, and only two K-tiles fit in front of them (two LDS buffers)
                wait_vmcnt<4>();
                if (k1_after) ep = 0;
            }
            VM_BAR();
            mma(1, 0, kt == 0);
            VM_BAR();
        }
        if (prestaged && nk < 2) ep = 0;
        prestaged = false;
        // Tile boundary: wave row 0 waits one barrier so that BOTH wave rows run their epilogues in the same interval
        // (their LDS round trips and store issue overlap instead of serialising); wave row 1 takes its matching extra
        // barrier after the epilogue, which also restores the one-barrier offset for the next tile.
        if (wr == 0) VM_BAR();
#ifdef VM_GEMM_ABLATE
        if (g.stamps && tid == 0 && tcount < 64) g.stamps[((size_t)blockIdx.x * 64 + tcount) * 2] = wall_clock64();
#endif
        // the consumed buffer is free (every wave's reads of it retired before this barrier): stage K-tile 1 of the
        // next tile into it NOW, ahead of the epilogue's stores in this CU's in-order memory pipe
        const bool do_prestage = has_next && nk >= 2;
        if (has_next) stage_bias(next_tile, (tcount + 1) & 1);
        if (do_prestage) {
            const int cb = (gk + 1) & 1;  // == buffer of the K-tile just consumed
            stage_Wa0(1, cb);
            stage_Xb0(1, cb);
            stage_Xb1(1, cb);
            stage_Wa1(1, cb);
        }
        // epilogue.  acc[i][j][e] = out[token t0 + wc*64 + 16j + r16][feature f0 + wr*128 + 16i + 4h + e].
        // 16-bit outputs go through a wave-private LDS transpose (the X region of the buffer just consumed: free
        // until phase 2 of the next K-tile) so that every store instruction writes 4 rows x 256 contiguous bytes
        // instead of 16 rows x 32 bytes.
        {
            int tm, tn;
            tile_coords(tile, tm, tn);
            const int t0 = tm << 8, f0 = tn << 8;
            if (ABL & 16) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
            } else if (OUT16) {
                // 2 KiB of wave-private scratch each in the spare LDS behind the bias slots: 8 token rows per pass
                const char *scratch = scratch_base + (wr * 4 + wc) * 2048;
                const float *bias_w = reinterpret_cast<const float *>(bias_slots + (tcount & 1) * 1024) + wr * 128;
                if (ABL & 8) epilogue16<DT, EPI, 4, true>(g, acc, bias_w, scratch, t0 + wc * 64, f0 + wr * 128, lane, tab_off);
                else epilogue16<DT, EPI, 4>(g, acc, bias_w, scratch, t0 + wc * 64, f0 + wr * 128, lane, tab_off);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = t0 + wc * 64 + j * 16 + r16;
                    f32x4 col[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) col[i] = acc[i][j];
                    if (t < M) epilogue_row<DT, EPI, 8>(g, col, t, f0 + wr * 128 + 4 * h);
                }
            }
        }
        if (wr == 1) VM_BAR();
#ifdef VM_GEMM_ABLATE
        if (g.stamps && tid == 0 && tcount < 64) g.stamps[((size_t)blockIdx.x * 64 + tcount) * 2 + 1] = wall_clock64();
#endif
        ++tcount;
        if (!has_next) break;
        {   // how many stores did this wave's epilogue leave in the pipe?  exact only for a full 16-bit tile
            int tm_done, tn_done;
            tile_coords(tile, tm_done, tn_done);
            const bool full = ((tm_done << 8) + 256 <= M) && OUT16 && !(ABL & 24);
            ep = full ? 1 : 2;
        }
        prestaged = do_prestage;
        if (!prestaged) {  // single-K-tile GEMMs: no counted skipping, drain everything once
            wait_vmcnt<0>();
            ep = 0;
        }
        if (!c0_start) clear_acc();
        tile = next_tile;
    }
#undef VM_WAIT_EP
    if (wr == 0) VM_BAR();  // match wave row 1's extra barrier
#undef VM_BAR
#undef VM_LGKM0
}

// ---------------------------------------------------------------------------------------------------------------
// 128 x 128 x 64, 4 waves (2 x 2), double-buffered, one barrier per K-tile
// ---------------------------------------------------------------------------------------------------------------
template <int DT, int EPI>
__global__ void __launch_bounds__(256, 2) gemm128_kernel(GemmArgs g) {
    using E = vm_elem<DT>;
    using vec8 = typename E::vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][W tile | X tile]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, h = lane >> 4;
    const int tiles_n = g.N >> 7;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int t0 = tm << 7, f0 = tn << 7;
    const int wf = wave & 1, wt = wave >> 1;
    const int K = g.K, M = g.M;
    constexpr bool TAB = EPI == EPI_GELU16 && !VM_GELU_POLY;
    if (TAB) gelu_tab_to_lds(g.gelu_tab, smem + 4 * HALF_BYTES, tid, 256);   // visible after the K loop's barriers
    const unsigned tab_off = lds_offset(smem + 4 * HALF_BYTES);

    const int srow = lane >> 3, scp = lane & 7;
    const uint16_t *wsrc[4], *xsrc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int row = wave * 32 + u * 8 + srow;
        const int chunk = scp ^ (row & 7);
        wsrc[u] = g.W + (size_t)(f0 + row) * K + chunk * 8;
        int tr = t0 + row;
        if (tr > M - 1) tr = M - 1;
        xsrc[u] = g.X + (size_t)tr * g.ldx + chunk * 8;
    }
    auto stage = [&](int kt, int buf) {
        char *wl = smem + buf * 2 * HALF_BYTES + wave * 32 * 128;
        char *xl = wl + HALF_BYTES;
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
        const char *wl = smem + buf * 2 * HALF_BYTES;
        const char *xl = wl + HALF_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            vec8 wf_[4], xf_[4];
            const int c = h + 4 * s;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int wrow = wf * 64 + i * 16 + r16;
                wf_[i] = *reinterpret_cast<const vec8 *>(wl + wrow * 128 + ((c ^ (wrow & 7)) << 4));
                const int xrow = wt * 64 + i * 16 + r16;
                xf_[i] = *reinterpret_cast<const vec8 *>(xl + xrow * 128 + ((c ^ (xrow & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = E::mfma16(wf_[i], xf_[j], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = t0 + wt * 64 + j * 16 + r16;
        if (t >= M) continue;
        f32x4 col[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) col[i] = acc[i][j];
        epilogue_row<DT, EPI, 4>(g, col, t, f0 + wf * 64 + 4 * h, tab_off);
    }
}

int g_variant = 0;  // 0 auto, 1 force 128^2, 2 force 256^2 one tile per block, 3 force 256^2 persistent (tools/gemm_bench, VIDMEM_GEMM)

template <int DT, int EPI>
int launch_epi(vm_ctx *ctx, const GemmArgs &g, hipStream_t st) {
    static const int env_variant = (int)VM_DEV_ENV("GEMM", 0);
    const int variant = g_variant ? g_variant : env_variant;
    constexpr size_t TAB_LDS = (EPI == EPI_GELU16 && !VM_GELU_POLY) ? VM_GELU_TAB_BYTES : 0;   // the GELU table in LDS
    const int tiles256 = ((g.M + 255) / 256) * (g.N / 256);
    // the 256 x 256 kernels address a tile through 32-bit byte offsets inside per-tile descriptors (gemm_guard.h)
    const bool big_ok = g.N % 256 == 0 && vm_gemm256_tile_addressable(g.K, g.ldx);
    // a 256^2 grid must give (nearly) every CU a tile; below that the 128^2 kernel fills the chip better
    const bool use256 = (variant == 2 || variant == 3) ? big_ok : (variant == 1 ? false : (big_ok && tiles256 * 10 >= ctx->num_cus * 8));
#ifdef VM_GEMM_ABLATE
    if (variant >= 1024 && DT == VM_F16) {  // persistent-kernel ablations: 1024 + ABL bits: 1024 + 8 (no stores) / + 16 (no epilogue)
        const size_t lds = 8 * HALF_BYTES + 2048 + 16384 + TAB_LDS;
        const int grid = tiles256 < ctx->num_cus ? tiles256 : ctx->num_cus;
#define ABLP(A)                                                                                                   \
    if (variant == 1024 + (A)) {                                                                                  \
        auto k = gemm256p_kernel<VM_F16, EPI, (A)>;                                                               \
        (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
        k<<<grid, 512, lds, st>>>(g);                                                                             \
        VM_LAUNCH_CHECK(ctx);                                                                                     \
        return VM_OK;                                                                                             \
    }
        ABLP(32) ABLP(64) ABLP(96) ABLP(96 + 16) ABLP(256)
#undef ABLP
        if (variant == 1024 + 8) {
            auto k = gemm256p_kernel<VM_F16, EPI, 8>;
            (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            k<<<grid, 512, lds, st>>>(g);
        } else {
            auto k = gemm256p_kernel<VM_F16, EPI, 16>;
            (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            k<<<grid, 512, lds, st>>>(g);
        }
        VM_LAUNCH_CHECK(ctx);
        return VM_OK;
    }
    if (variant >= 16 && EPI == EPI_STORE16 && DT == VM_F16) {
        const size_t lds = 8 * HALF_BYTES;
        const int abl = variant >> 4;
#define ABLGO(A)                                                                                                  \
    case A: {                                                                                                     \
        auto k = gemm256_kernel<VM_F16, EPI_STORE16, A>;                                                          \
        (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
        k<<<tiles256, 512, lds, st>>>(g);                                                                         \
    } break;
        switch (abl) { ABLGO(1) ABLGO(2) ABLGO(3) ABLGO(4) ABLGO(5) ABLGO(6) ABLGO(7) ABLGO(8) default: break; }
#undef ABLGO
        VM_LAUNCH_CHECK(ctx);
        return VM_OK;
    }
#endif
    if (use256 && variant != 2 && g.N <= 4096) {
        auto kern = gemm256p_kernel<DT, EPI>;
        static unsigned long long attr_set_p = 0;   // one bit per device
        const size_t lds = 8 * HALF_BYTES + 2048 + 16384 + TAB_LDS;  // staging + bias slots + epilogue scratch (+ table)
        if (!((attr_set_p >> (ctx->device & 63)) & 1ull)) {
            VM_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            163840));
            attr_set_p |= 1ull << (ctx->device & 63);
        }
        const int grid = tiles256 < ctx->num_cus ? tiles256 : ctx->num_cus;
        kern<<<grid, 512, lds, st>>>(g);
    } else if (use256) {
        auto kern = gemm256_kernel<DT, EPI>;
        static unsigned long long attr_set = 0;   // one bit per device
        const size_t lds = 8 * HALF_BYTES + TAB_LDS;
        if (!((attr_set >> (ctx->device & 63)) & 1ull)) {
            VM_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set |= 1ull << (ctx->device & 63);
        }
        kern<<<tiles256, 512, lds, st>>>(g);
    } else {
        auto kern = gemm128_kernel<DT, EPI>;
        static unsigned long long attr_set = 0;   // one bit per device
        const size_t lds = 4 * HALF_BYTES + TAB_LDS;
        if (!((attr_set >> (ctx->device & 63)) & 1ull)) {
            VM_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set |= 1ull << (ctx->device & 63);
        }
        kern<<<((g.M + 127) / 128) * (g.N / 128), 256, lds, st>>>(g);
    }
    VM_LAUNCH_CHECK(ctx);
    return VM_OK;
}

template <int DT>
int launch(vm_ctx *ctx, const GemmArgs &g, int epi, hipStream_t st) {
    vm_prof_scope prof(ctx, g.prof_cat, st);
    switch (epi) {
        case EPI_STORE16: return launch_epi<DT, EPI_STORE16>(ctx, g, st);
        case EPI_DELTA16:  // identical to STORE16 for an fp16 encoder: no second instantiation
            if (DT == VM_F16) return launch_epi<DT, EPI_STORE16>(ctx, g, st);
            return launch_epi<DT, EPI_DELTA16>(ctx, g, st);
        case EPI_GELU16: return launch_epi<DT, EPI_GELU16>(ctx, g, st);
        case EPI_QGELU16: return launch_epi<DT, EPI_QGELU16>(ctx, g, st);
        case EPI_RESID32: return launch_epi<DT, EPI_RESID32>(ctx, g, st);
        case EPI_PATCH: return launch_epi<DT, EPI_PATCH>(ctx, g, st);
        default: return vm_fail(ctx, VM_ERR_INVALID, "bad epilogue %d", epi);
    }
}

}  // namespace

void vm_gemm_set_variant(int v) { g_variant = v; }

int vm_gemm(vm_ctx *ctx, int dtype, const GemmArgs &g, int epi, hipStream_t st) {
    if (g.M <= 0 || g.N % 128 != 0 || g.K % BK != 0 || g.K <= 0)
        return vm_fail(ctx, VM_ERR_UNSUPPORTED, "gemm shape M=%d N=%d K=%d (need N%%128==0, K%%64==0)", g.M, g.N,
                       g.K);
    GemmArgs a = g;
    a.stream_out = (size_t)g.M * g.N * 2 > ((size_t)32 << 20);  // more than the 8 x 4 MiB of L2
    if (a.hm_rows <= 0) a.hm_rows = g.M;
    if (a.hm_stride <= 0) a.hm_stride = 1;
    a.gelu_tab = ctx ? ctx->gelu_tab : nullptr;
    if (epi == EPI_GELU16 && !a.gelu_tab) return vm_fail(ctx, VM_ERR_INVALID, "vm_gemm: GELU epilogue needs a context");
    {   // feature-tile groups of the persistent 256 x 256 kernel (gemm256p_kernel, "Tile order")
        // weight bytes an XCD's L2 keeps beside the streams; 0 = off
        static const long budget = VM_DEV_ENV("GEMM_WGROUP_KB", 2560) * 1024;
        a.fgroup = 0;
        static const int zero_env = (int)VM_DEV_ENV("GEMM_ZERO", 0);
        a.explicit_zero = zero_env;
        const int tiles_n = g.N / 256;
        const long wtile = 256L * g.K * 2, wall = wtile * tiles_n;
        if (budget > 0 && g.N % 256 == 0 && wall > budget && wtile <= budget) {
            const int ngroups = (int)((wall + budget - 1) / budget);
            const int fg = (tiles_n + ngroups - 1) / ngroups;
            const int ng = (tiles_n + fg - 1) / fg;
            // worth it only when re-streaming the activations costs less than re-fetching the weights every round
            const double x_extra = (double)(ng - 1) * g.M * g.K * 2;
            const double rounds = (double)((g.M + 255) / 256) * tiles_n / (ctx ? ctx->num_cus : 256);
            const double w_refetch = rounds * 8.0 * wall;
            if (x_extra < w_refetch) a.fgroup = fg;
        }
    }
    return dtype == VM_F16 ? launch<VM_F16>(ctx, a, epi, st) : launch<VM_BF16>(ctx, a, epi, st);
}
