// Exhaustive fp64 cosine: every (query, row) pair scored exactly as the reference does
// (src/components/pre_llm_injector.py:374-388: three left-to-right fp64 sums, sqrt, guards, one division),
// then a stable top-k.  Used (a) as the fallback for queries the fast scan cannot certify, (b) for the
// post-compression filter (src/pipeline/retriever_hybrid.py:494-504), (c) as an on-device checker.
#include "vm_internal.h"

namespace {

// grid (ceil(n/256), Q).  Thread = one stored row; the query sits in LDS.  `order_to_phys`: rows are addressed
// by age order o (0 = oldest) and mapped to the physical slot (o + head) % cap, so out[q, o] is in row-id order.
template <int DT>
__global__ void __launch_bounds__(256)
    cosine_exact_kernel(const uint16_t *__restrict__ queries, const uint16_t *__restrict__ rows, int64_t n,
                        int64_t head, int64_t cap, int D, double *__restrict__ out, int64_t out_stride) {
    using E = vm_elem<DT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t *ql = reinterpret_cast<uint16_t *>(smem);
    __shared__ double qnorm_sh;
    const int q = blockIdx.y;
    const uint16_t *qv = queries + (size_t)q * D;
    for (int i = threadIdx.x; i < D / 8; i += blockDim.x)
        reinterpret_cast<uint4 *>(ql)[i] = reinterpret_cast<const uint4 *>(qv)[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        double nq = 0.0;
        for (int i = 0; i < D; ++i) {
            const double x = E::to_double(ql[i]);
            nq = __dadd_rn(nq, __dmul_rn(x, x));
        }
        qnorm_sh = __dsqrt_rn(nq);
    }
    __syncthreads();
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n) return;
    int64_t p = o + head;
    if (p >= cap) p -= cap;
    const uint16_t *mv = rows + (size_t)p * D;
    double dot = 0.0, nb = 0.0;
    for (int i = 0; i < D; i += 8) {
        const uint4 b = *reinterpret_cast<const uint4 *>(mv + i);
        const uint4 a = *reinterpret_cast<const uint4 *>(ql + i);
        const uint16_t *ae = reinterpret_cast<const uint16_t *>(&a);
        const uint16_t *be = reinterpret_cast<const uint16_t *>(&b);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double x = E::to_double(ae[j]), y = E::to_double(be[j]);
            dot = __dadd_rn(dot, __dmul_rn(x, y));
            nb = __dadd_rn(nb, __dmul_rn(y, y));
        }
    }
    const double qn = qnorm_sh, mn = __dsqrt_rn(nb);
    out[(size_t)q * out_stride + o] = (qn == 0.0 || mn == 0.0) ? 0.0 : __ddiv_rn(dot, __dmul_rn(qn, mn));
}

// One block per query: k rounds, each finds the best key strictly after the previous winner in
// (score desc, index asc) order.  Stateless, so any n works.
__global__ void __launch_bounds__(256)
    topk_select_kernel(const double *__restrict__ scores, int64_t n, int64_t stride, int k, int use_min,
                       double min_score, int score_mode, int64_t base, int64_t row_stride, int64_t row_offset,
                       double *__restrict__ out_scores, int64_t *__restrict__ out_rows) {
    __shared__ double ws[256];
    __shared__ int64_t wi[256];
    const int q = blockIdx.x, tid = threadIdx.x;
    const double *s = scores + (size_t)q * stride;
    double prev_s = INFINITY;
    int64_t prev_i = -1;
    for (int r = 0; r < k; ++r) {
        double bs = -INFINITY;
        int64_t bi = -1;
        for (int64_t i = tid; i < n; i += 256) {
            const double v = s[i];
            const bool after_prev = v < prev_s || (v == prev_s && i > prev_i);
            const bool beats = bi < 0 || v > bs || (v == bs && i < bi);
            if (after_prev && beats) {
                bs = v;
                bi = i;
            }
        }
        ws[tid] = bs;
        wi[tid] = bi;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off) {
                const double s2 = ws[tid + off];
                const int64_t i2 = wi[tid + off];
                const bool take = i2 >= 0 && (wi[tid] < 0 || s2 > ws[tid] || (s2 == ws[tid] && i2 < wi[tid]));
                if (take) {
                    ws[tid] = s2;
                    wi[tid] = i2;
                }
            }
            __syncthreads();
        }
        prev_s = ws[0];
        prev_i = wi[0];
        __syncthreads();
        if (tid == 0) {
            double shown = prev_s;
            bool ok = prev_i >= 0;
            if (ok) {
                if (score_mode == VM_SCORE_UNIT_INTERVAL) shown = __ddiv_rn(__dadd_rn(1.0, prev_s), 2.0);
                if (use_min && !(shown > min_score)) ok = false;
            }
            out_scores[(size_t)q * k + r] = ok ? shown : 0.0;
            out_rows[(size_t)q * k + r] = ok ? (base + prev_i) * row_stride + row_offset : -1;
        }
        if (prev_i < 0) {  // exhausted: pad the rest
            for (int r2 = r + 1 + tid; r2 < k; r2 += 256) {
                out_scores[(size_t)q * k + r2] = 0.0;
                out_rows[(size_t)q * k + r2] = -1;
            }
            break;
        }
    }
}

}  // namespace

extern "C" int vm_cosine_exact(vm_ctx *ctx, const void *queries, int Q, const void *rows, int64_t S, int D,
                               int dtype, double *out, void *stream) {
    if (!ctx || !queries || !out || Q <= 0 || S < 0 || (S > 0 && !rows))
        return vm_fail(ctx, VM_ERR_INVALID, "vm_cosine_exact: bad arguments");
    if (D <= 0 || D % 8 != 0) return vm_fail(ctx, VM_ERR_UNSUPPORTED, "vm_cosine_exact: D=%d not a multiple of 8", D);
    if (S == 0) return VM_OK;
    dim3 grid((unsigned)((S + 255) / 256), Q);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VM_F16)
        cosine_exact_kernel<VM_F16><<<grid, 256, (size_t)D * 2, st>>>((const uint16_t *)queries,
                                                                     (const uint16_t *)rows, S, 0, S, D, out, S);
    else if (dtype == VM_BF16)
        cosine_exact_kernel<VM_BF16><<<grid, 256, (size_t)D * 2, st>>>((const uint16_t *)queries,
                                                                      (const uint16_t *)rows, S, 0, S, D, out, S);
    else
        return vm_fail(ctx, VM_ERR_INVALID, "bad dtype %d", dtype);
    VM_LAUNCH_CHECK(ctx);
    return VM_OK;
}

extern "C" size_t vm_topk_exact_workspace_bytes(const vm_memory *m, int Q, int k) {
    (void)k;
    if (!m || Q <= 0) return 0;
    return vm_align_up((size_t)Q * (size_t)m->cap * 8, 256);
}

extern "C" int vm_topk_cosine_exact(vm_memory *m, const void *queries, int Q, int k, int use_min_score,
                                    double min_score, int score_mode, int64_t row_stride, int64_t row_offset,
                                    double *out_scores, int64_t *out_rows, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    if (!m) return VM_ERR_INVALID;
    vm_ctx *ctx = m->ctx;
    if (!queries || !out_scores || !out_rows || Q <= 0 || k <= 0)
        return vm_fail(ctx, VM_ERR_INVALID, "vm_topk_cosine_exact: bad arguments");
    // This entry point reads the HOST mirror of the row count (it sizes the grid), so it is not for graph replay.
    const int64_t total = m->h_total;
    int64_t n = total < m->cap ? total : m->cap, head = 0, base = 0;
    if (m->ring && total > m->cap) {
        head = total % m->cap;
        base = total - m->cap;
    }
    if (!workspace || workspace_bytes < (size_t)Q * (size_t)(n > 0 ? n : 1) * 8)
        return vm_fail(ctx, VM_ERR_NOMEM, "vm_topk_cosine_exact: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    vm_prof_scope prof(ctx, VM_PROF_TOPK_EXACT, st);
    double *sc = (double *)workspace;
    if (n > 0) {
        dim3 grid((unsigned)((n + 255) / 256), Q);
        if (m->dtype == VM_F16)
            cosine_exact_kernel<VM_F16><<<grid, 256, (size_t)m->D * 2, st>>>((const uint16_t *)queries, m->rows, n,
                                                                            head, m->cap, m->D, sc, n);
        else
            cosine_exact_kernel<VM_BF16><<<grid, 256, (size_t)m->D * 2, st>>>((const uint16_t *)queries, m->rows,
                                                                             n, head, m->cap, m->D, sc, n);
        VM_LAUNCH_CHECK(ctx);
    }
    topk_select_kernel<<<Q, 256, 0, st>>>(sc, n, n, k, use_min_score, min_score, score_mode, base, row_stride,
                                          row_offset, out_scores, out_rows);
    VM_LAUNCH_CHECK(ctx);
    return VM_OK;
}
