// Internal definitions shared by the libvidmem translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/vidmem.h"

// Developer A/B switches.  The release library (make / __graft_entry__.build()) compiles every one of them to its
// default: no environment variable selects which kernel produces product results, and no VIDMEM_* name appears in
// libvidmem.so.  tools/dev_build.sh builds with -DVM_DEV_SWITCHES, where each is read from the environment once per
// process.  What a deployment may choose is an explicit per-handle option (vm_encoder_set_option, include/vidmem.h).
#ifdef VM_DEV_SWITCHES
static inline long vm_dev_env_(const char *name, long dflt) {
    const char *e = getenv(name);
    return e ? atol(e) : dflt;
}
#define VM_DEV_ENV(name, dflt) vm_dev_env_("VIDMEM_" name, (dflt))
#else
#define VM_DEV_ENV(name, dflt) ((long)(dflt))
#endif

struct vm_ctx {
    int device;
    int num_cus;
    char err[512];
    // optional per-kernel timing (vm_profile_enable): event pool, 2 events + 1 category per launch
    hipEvent_t *prof_ev;
    int *prof_cat;
    int prof_cap, prof_n;
    uint32_t prof_mask;
    double prof_ms[VM_PROF_NCAT];
    int64_t prof_launches[VM_PROF_NCAT];
    // erf-GELU table of the FC1 epilogue (gemm.hip gelu_tab_*; built in vm_init, context.hip): VM_GELU_TAB_N pairs
    // {a, b} with Phi(x) ~ a + b x on the 1/128-wide interval around x = -5 + i/128, device memory
    float *gelu_tab;
};

constexpr int VM_GELU_TAB_N = 1281;                 // x = -5 ... +5 in steps of 1/128
constexpr int VM_GELU_TAB_BYTES = 10256;            // 1281 x 8 B, padded to whole 16-byte pieces

// Brackets the launches inside a scope with two events when profiling is on; free when it is off.
struct vm_prof_scope {
    vm_ctx *ctx;
    hipStream_t st;
    int slot;
    vm_prof_scope(vm_ctx *c, int cat, hipStream_t s) : ctx(c), st(s), slot(-1) {
        if (c && c->prof_ev && c->prof_n < c->prof_cap && ((c->prof_mask >> cat) & 1u)) {
            slot = c->prof_n++;
            c->prof_cat[slot] = cat;
            (void)hipEventRecord(c->prof_ev[2 * slot], s);
        }
    }
    ~vm_prof_scope() {
        if (slot >= 0) (void)hipEventRecord(ctx->prof_ev[2 * slot + 1], st);
    }
};

inline int vm_fail(vm_ctx *ctx, int code, const char *fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define VM_HIP(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return vm_fail((ctx), VM_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),   \
                           __FILE__, __LINE__);                                                        \
    } while (0)

#define VM_LAUNCH_CHECK(ctx)                                                                           \
    do {                                                                                               \
        hipError_t e_ = hipGetLastError();                                                             \
        if (e_ != hipSuccess)                                                                          \
            return vm_fail((ctx), VM_ERR_HIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), \
                           __FILE__, __LINE__);                                                        \
    } while (0)

static inline size_t vm_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- 16-bit float helpers (device) ------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int DT>
struct vm_elem;  // DT = VM_F16 / VM_BF16
template <>
struct vm_elem<VM_F16> {
    using vec8 = f16x8;
    static __device__ __forceinline__ float to_float(uint16_t b) {
        _Float16 h;
        __builtin_memcpy(&h, &b, 2);
        return (float)h;
    }
    static __device__ __forceinline__ double to_double(uint16_t b) { return (double)to_float(b); }
    static __device__ __forceinline__ uint16_t from_float(float f) {
        _Float16 h = (_Float16)f;  // round-to-nearest-even
        uint16_t b;
        __builtin_memcpy(&b, &h, 2);
        return b;
    }
    // two values -> one packed dword (lo = a): a single v_cvt_pk_f16_f32, same RNE rounding as from_float
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        const h2 v = __builtin_convertvector(f2{a, b}, h2);
        uint32_t u;
        __builtin_memcpy(&u, &v, 4);
        return u;
    }
    static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <>
struct vm_elem<VM_BF16> {
    using vec8 = bf16x8;
    static __device__ __forceinline__ float to_float(uint16_t b) {
        uint32_t u = ((uint32_t)b) << 16;
        return __builtin_bit_cast(float, u);
    }
    static __device__ __forceinline__ double to_double(uint16_t b) { return (double)to_float(b); }
    static __device__ __forceinline__ uint16_t from_float(float f) {
        __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
        uint16_t b;
        __builtin_memcpy(&b, &h, 2);
        return b;
    }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {  // v_cvt_pk_bf16_f32
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        const b2 v = __builtin_convertvector(f2{a, b}, b2);
        uint32_t u;
        __builtin_memcpy(&u, &v, 4);
        return u;
    }
    static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
