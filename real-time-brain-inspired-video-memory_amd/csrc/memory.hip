// Embedding memory: resident [capacity, D] 16-bit rows + exact fp64 norms + fp32 reciprocal norms.
// Replaces the Chunk.embedding property store of the reference
// (src/components/neo4j_handler.py:229-242 append, src/components/pre_llm_injector.py:390-412 read-back).
#include "vm_internal.h"

// One block per appended row: coalesced 16-B copy of the row into its slot, then lane 0 accumulates the
// reference's norm exactly: sqrt(sum(b*b)) with one rounding per product and per partial sum, left to right
// (src/components/pre_llm_injector.py:383).  The slot comes from the DEVICE-side counter so a captured graph
// replays correctly.
template <int DT>
__global__ void __launch_bounds__(128) memory_append_kernel(const uint16_t *__restrict__ src, int B, int D,
                                                            uint16_t *__restrict__ rows,
                                                            double *__restrict__ norm64,
                                                            float *__restrict__ rnorm32,
                                                            const int64_t *__restrict__ d_total, int64_t cap,
                                                            int ring) {
    const int b = blockIdx.x;
    const int64_t total = *d_total;
    int64_t id = total + b;
    int64_t slot = ring ? (id % cap) : id;
    if (slot >= cap) return;  // non-ring overflow is rejected on the host; never write out of bounds
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src + (size_t)b * D);
    uint4 *d4 = reinterpret_cast<uint4 *>(rows + (size_t)slot * D);
    for (int i = threadIdx.x; i < D / 8; i += blockDim.x) d4[i] = s4[i];
    if (threadIdx.x == 0) {
        const uint16_t *s = src + (size_t)b * D;
        double acc = 0.0;
        for (int i = 0; i < D; i += 8) {
            uint4 v = *reinterpret_cast<const uint4 *>(s + i);
            const uint16_t *e = reinterpret_cast<const uint16_t *>(&v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                double x = vm_elem<DT>::to_double(e[j]);
                acc = __dadd_rn(acc, __dmul_rn(x, x));
            }
        }
        double nrm = __dsqrt_rn(acc);
        norm64[slot] = nrm;
        rnorm32[slot] = nrm > 0.0 ? (float)(1.0 / nrm) : 0.0f;
    }
}

__global__ void memory_bump_kernel(int64_t *d_total, int B) { *d_total += B; }

extern "C" int vm_memory_create(vm_ctx *ctx, int64_t capacity_rows, int D, int dtype, int ring,
                                vm_memory **out) {
    if (!ctx || !out) return VM_ERR_INVALID;
    if (capacity_rows <= 0 || capacity_rows >= (int64_t)0x7fffff00)
        return vm_fail(ctx, VM_ERR_INVALID, "capacity_rows %lld out of range", (long long)capacity_rows);
    if (D <= 0 || D % 128 != 0)
        return vm_fail(ctx, VM_ERR_UNSUPPORTED, "D=%d: embedding dimension must be a multiple of 128", D);
    if (dtype != VM_F16 && dtype != VM_BF16) return vm_fail(ctx, VM_ERR_INVALID, "bad dtype %d", dtype);
    VM_HIP(ctx, hipSetDevice(ctx->device));
    vm_memory *m = new vm_memory();
    memset(m, 0, sizeof(*m));
    m->ctx = ctx;
    m->cap = capacity_rows;
    m->D = D;
    m->dtype = dtype;
    m->ring = ring ? 1 : 0;
    const int64_t cap_pad = (capacity_rows + 63) / 64 * 64;  // tail tiles may read (never use) past cap
    hipError_t e = hipMalloc((void **)&m->rows, (size_t)cap_pad * D * 2);
    if (e == hipSuccess) e = hipMalloc((void **)&m->norm64, (size_t)cap_pad * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&m->rnorm32, (size_t)cap_pad * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_total, 64);
    if (e != hipSuccess) {
        vm_memory_destroy(m);
        return vm_fail(ctx, VM_ERR_NOMEM, "hipMalloc for %lld x %d memory failed: %s", (long long)capacity_rows,
                       D, hipGetErrorString(e));
    }
    VM_HIP(ctx, hipMemset(m->rows, 0, (size_t)cap_pad * D * 2));
    VM_HIP(ctx, hipMemset(m->norm64, 0, (size_t)cap_pad * 8));
    VM_HIP(ctx, hipMemset(m->rnorm32, 0, (size_t)cap_pad * 4));
    VM_HIP(ctx, hipMemset(m->d_total, 0, 64));
    *out = m;
    return VM_OK;
}

extern "C" void vm_memory_destroy(vm_memory *m) {
    if (!m) return;
    if (m->rows) (void)hipFree(m->rows);
    if (m->norm64) (void)hipFree(m->norm64);
    if (m->rnorm32) (void)hipFree(m->rnorm32);
    if (m->d_total) (void)hipFree(m->d_total);
    delete m;
}

extern "C" int vm_memory_append(vm_memory *m, const void *rows, int B, int64_t *out_first_row_host,
                                void *stream) {
    if (!m) return VM_ERR_INVALID;
    vm_ctx *ctx = m->ctx;
    if (B < 0 || (B > 0 && !rows)) return vm_fail(ctx, VM_ERR_INVALID, "vm_memory_append: bad arguments");
    if (out_first_row_host) *out_first_row_host = m->h_total;
    if (B == 0) return VM_OK;
    if (!m->ring && m->h_total + B > m->cap)
        return vm_fail(ctx, VM_ERR_NOMEM, "memory full: %lld + %d > capacity %lld", (long long)m->h_total, B,
                       (long long)m->cap);
    if (m->ring && B > m->cap) return vm_fail(ctx, VM_ERR_INVALID, "append of %d rows exceeds ring capacity", B);
    hipStream_t st = (hipStream_t)stream;
    vm_prof_scope prof(ctx, VM_PROF_APPEND, st);
    if (m->dtype == VM_F16)
        memory_append_kernel<VM_F16><<<B, 128, 0, st>>>((const uint16_t *)rows, B, m->D, m->rows, m->norm64,
                                                       m->rnorm32, m->d_total, m->cap, m->ring);
    else
        memory_append_kernel<VM_BF16><<<B, 128, 0, st>>>((const uint16_t *)rows, B, m->D, m->rows, m->norm64,
                                                        m->rnorm32, m->d_total, m->cap, m->ring);
    VM_LAUNCH_CHECK(ctx);
    memory_bump_kernel<<<1, 1, 0, st>>>(m->d_total, B);
    VM_LAUNCH_CHECK(ctx);
    m->h_total += B;
    return VM_OK;
}

extern "C" int64_t vm_memory_size(const vm_memory *m) { return m ? m->h_total : 0; }
extern "C" int64_t vm_memory_capacity(const vm_memory *m) { return m ? m->cap : 0; }
extern "C" int vm_memory_dim(const vm_memory *m) { return m ? m->D : 0; }
extern "C" const void *vm_memory_rows(const vm_memory *m) { return m ? m->rows : nullptr; }

extern "C" int64_t vm_memory_sync(vm_memory *m, void *stream) {
    if (!m) return VM_ERR_INVALID;
    int64_t total = 0;
    VM_HIP(m->ctx, hipMemcpyAsync(&total, m->d_total, sizeof(total), hipMemcpyDeviceToHost, (hipStream_t)stream));
    VM_HIP(m->ctx, hipStreamSynchronize((hipStream_t)stream));
    m->h_total = total;
    return total;
}

extern "C" int vm_memory_reset(vm_memory *m, void *stream) {
    if (!m) return VM_ERR_INVALID;
    VM_HIP(m->ctx, hipMemsetAsync(m->d_total, 0, 64, (hipStream_t)stream));
    m->h_total = 0;
    return VM_OK;
}
