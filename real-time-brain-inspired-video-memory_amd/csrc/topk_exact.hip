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

// fp32 operands (vm_cosine_exact, VM_F32): same arithmetic on the values as they are - every fp32 is exact in fp64.
// grid (ceil(n/256), Q); S is small here (the segments of a handful of hits), so the query stays in global memory.
__global__ void __launch_bounds__(256)
    cosine_exact_f32_kernel(const float *__restrict__ queries, const float *__restrict__ rows, int64_t n, int D,
                            double *__restrict__ out) {
    const int q = blockIdx.y;
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n) return;
    const float *qv = queries + (size_t)q * D, *mv = rows + (size_t)o * D;
    double dot = 0.0, na = 0.0, nb = 0.0;
    for (int i = 0; i < D; ++i) {
        const double x = (double)qv[i], y = (double)mv[i];
        dot = __dadd_rn(dot, __dmul_rn(x, y));
        na = __dadd_rn(na, __dmul_rn(x, x));
        nb = __dadd_rn(nb, __dmul_rn(y, y));
    }
    const double qn = __dsqrt_rn(na), mn = __dsqrt_rn(nb);
    out[(size_t)q * n + o] = (qn == 0.0 || mn == 0.0) ? 0.0 : __ddiv_rn(dot, __dmul_rn(qn, mn));
}

// One block per query: k rounds, each finds the best key strictly after the previous winner in
// (score desc, index asc) order.  Stateless, so any n works.  col_limit (may be null): query q sees only the first
// min(n, col_limit[q]) columns (vm_topk_select: the rows that existed before the query's own chunk was appended).
__global__ void __launch_bounds__(256)
    topk_select_kernel(const double *__restrict__ scores, int64_t n, int64_t stride, int k, int use_min,
                       double min_score, int score_mode, int64_t base, int64_t row_stride, int64_t row_offset,
                       double *__restrict__ out_scores, int64_t *__restrict__ out_rows,
                       const int64_t *__restrict__ col_limit) {
    __shared__ double ws[256];
    __shared__ int64_t wi[256];
    const int q = blockIdx.x, tid = threadIdx.x;
    const double *s = scores + (size_t)q * stride;
    if (col_limit) {
        const int64_t lim = col_limit[q];
        n = lim < 0 ? 0 : (lim < n ? lim : n);
    }
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

// ---------------------------------------------------------------------------------------------------------
// Exhaustive redo of the queries the fast scan could not certify (vm_topk_cosine's out_query_flags).
// Both kernels read the row count from the DEVICE counter and take flags from device memory: no host read-back,
// no allocation, capturable into a hipGraph, and free (two near-empty launches) when no query is flagged.
// ---------------------------------------------------------------------------------------------------------
constexpr int REDO_THREADS = 256;
constexpr int REDO_CHUNK = 2048;  // scores held in LDS per selection pass
constexpr int REDO_KMAX = 64;

// k rounds of block-wide arg-best over `n` (score, order) candidates read through `get`, strictly after the
// previous winner in (score desc, order asc): a stable top-k without sorting.  Winners go to out_s / out_o (LDS),
// -inf / -1 padded.  All threads of the block must call it.
template <typename Get>
__device__ __forceinline__ void block_select(int n, int k, Get get, double *out_s, int64_t *out_o, double *red_s,
                                             int64_t *red_o) {
    const int tid = threadIdx.x;
    double prev_s = INFINITY;
    int64_t prev_o = -1;
    for (int r = 0; r < k; ++r) {
        double bs = -INFINITY;
        int64_t bo = -1;
        for (int i = tid; i < n; i += REDO_THREADS) {
            double v;
            int64_t o;
            get(i, v, o);
            if (o < 0) continue;
            const bool after_prev = v < prev_s || (v == prev_s && o > prev_o);
            const bool beats = bo < 0 || v > bs || (v == bs && o < bo);
            if (after_prev && beats) {
                bs = v;
                bo = o;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double s2 = __shfl_xor(bs, off, 64);
            const int64_t o2 = __shfl_xor(bo, off, 64);
            if (o2 >= 0 && (bo < 0 || s2 > bs || (s2 == bs && o2 < bo))) {
                bs = s2;
                bo = o2;
            }
        }
        __syncthreads();  // previous round's readers of red_* are done
        if ((tid & 63) == 0) {
            red_s[tid >> 6] = bs;
            red_o[tid >> 6] = bo;
        }
        __syncthreads();
        bs = red_s[0];
        bo = red_o[0];
#pragma unroll
        for (int w = 1; w < REDO_THREADS / 64; ++w) {
            const double s2 = red_s[w];
            const int64_t o2 = red_o[w];
            if (o2 >= 0 && (bo < 0 || s2 > bs || (s2 == bs && o2 < bo))) {
                bs = s2;
                bo = o2;
            }
        }
        if (tid == 0) {
            out_s[r] = bo >= 0 ? bs : -INFINITY;
            out_o[r] = bo;
        }
        prev_s = bs;
        prev_o = bo;
        if (bo < 0) {  // exhausted (uniform): pad the rest
            for (int r2 = r + 1 + tid; r2 < k; r2 += REDO_THREADS) {
                out_s[r2] = -INFINITY;
                out_o[r2] = -1;
            }
            break;
        }
    }
    __syncthreads();
}

// grid = nblk row blocks; every block walks all Q flags and, for each flagged query, scores its contiguous slice of
// age orders exactly as the reference does (src/components/pre_llm_injector.py:374-388) and keeps the slice's
// stable top-k: part[(block * Q + q) * k + i] = {score fp64, age order int64}.
template <int DT>
__global__ void __launch_bounds__(REDO_THREADS)
    topk_redo_scan_kernel(const uint16_t *__restrict__ queries, const uint16_t *__restrict__ rows,
                          const double *__restrict__ norm64, const int64_t *__restrict__ d_total, int64_t cap,
                          int ring, int D, int Q, int k, const int32_t *__restrict__ flags,
                          double *__restrict__ part_s, int64_t *__restrict__ part_o) {
    using E = vm_elem<DT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t *ql = reinterpret_cast<uint16_t *>(smem);                       // [D]
    double *sc = reinterpret_cast<double *>(smem + (size_t)D * 2);           // [REDO_CHUNK]
    __shared__ double run_s[REDO_KMAX], new_s[REDO_KMAX], red_s[REDO_THREADS / 64];
    __shared__ int64_t run_o[REDO_KMAX], new_o[REDO_KMAX], red_o[REDO_THREADS / 64];
    __shared__ double qnorm_sh;
    const int tid = threadIdx.x;
    const RingView rv = ring_view(*d_total, cap, ring);
    const int64_t per = (rv.n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per;
    const int64_t hi = lo + per < rv.n ? lo + per : rv.n;
    // The common launch has no flagged query at all: find that out with independent strided loads and one block-wide OR
    // (the per-query test below is a chain of dependent scalar loads: 0.43 ms for the 7,040 queries an 8-GPU step
    // brings to every shard).
    int any = 0;
    for (int i = tid; i < Q; i += REDO_THREADS) any |= flags[i];
    if (!__syncthreads_or(any)) return;
    for (int q = 0; q < Q; ++q) {
        if (flags[q] == 0) continue;  // uniform
        __syncthreads();
        for (int i = tid; i < D / 8; i += REDO_THREADS)
            reinterpret_cast<uint4 *>(ql)[i] = reinterpret_cast<const uint4 *>(queries + (size_t)q * D)[i];
        if (tid < k) {
            run_s[tid] = -INFINITY;
            run_o[tid] = -1;
        }
        __syncthreads();
        if (tid == 0) {
            double nq = 0.0;
            for (int i = 0; i < D; ++i) {
                const double x = E::to_double(ql[i]);
                nq = __dadd_rn(nq, __dmul_rn(x, x));
            }
            qnorm_sh = __dsqrt_rn(nq);
        }
        __syncthreads();
        const double qn = qnorm_sh;
        for (int64_t c0 = lo; c0 < hi; c0 += REDO_CHUNK) {
            const int cn = (int)(hi - c0 < REDO_CHUNK ? hi - c0 : REDO_CHUNK);
            for (int i = tid; i < cn; i += REDO_THREADS) {
                int64_t p = c0 + i + rv.head;
                if (p >= rv.cap) p -= rv.cap;
                const uint16_t *mv = rows + (size_t)p * D;
                double dot = 0.0;
                for (int e0 = 0; e0 < D; e0 += 8) {
                    const uint4 b = *reinterpret_cast<const uint4 *>(mv + e0);
                    const uint4 a = *reinterpret_cast<const uint4 *>(ql + e0);
                    const uint16_t *ae = reinterpret_cast<const uint16_t *>(&a);
                    const uint16_t *be = reinterpret_cast<const uint16_t *>(&b);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        dot = __dadd_rn(dot, __dmul_rn(E::to_double(ae[j]), E::to_double(be[j])));
                }
                const double mn = norm64[p];  // the reference's norm of the stored row, computed at append
                sc[i] = (qn == 0.0 || mn == 0.0) ? 0.0 : __ddiv_rn(dot, __dmul_rn(qn, mn));
            }
            __syncthreads();
            // candidates = this chunk's scores followed by the running list
            block_select(cn + k, k,
                         [&](int i, double &v, int64_t &o) {
                             if (i < cn) {
                                 v = sc[i];
                                 o = c0 + i;
                             } else {
                                 v = run_s[i - cn];
                                 o = run_o[i - cn];
                             }
                         },
                         new_s, new_o, red_s, red_o);
            if (tid < k) {
                run_s[tid] = new_s[tid];
                run_o[tid] = new_o[tid];
            }
            __syncthreads();
        }
        if (tid < k) {
            part_s[((size_t)blockIdx.x * Q + q) * k + tid] = run_s[tid];
            part_o[((size_t)blockIdx.x * Q + q) * k + tid] = run_o[tid];
        }
    }
}

// grid = Q; a block whose query is not flagged exits at once.  Stable top-k over the nblk * k slice winners,
// then the same output mapping as the fast path (score mode, min_score, global row id, -1 / 0.0 padding).
__global__ void __launch_bounds__(REDO_THREADS)
    topk_redo_merge_kernel(const double *__restrict__ part_s, const int64_t *__restrict__ part_o, int nblk, int Q,
                           int k, const int32_t *__restrict__ flags, const int64_t *__restrict__ d_total,
                           int64_t cap, int ring, int use_min, double min_score, int score_mode,
                           int64_t row_stride, int64_t row_offset, double *__restrict__ out_scores,
                           int64_t *__restrict__ out_rows) {
    __shared__ double win_s[REDO_KMAX], red_s[REDO_THREADS / 64];
    __shared__ int64_t win_o[REDO_KMAX], red_o[REDO_THREADS / 64];
    const int q = blockIdx.x, tid = threadIdx.x;
    if (flags[q] == 0) return;
    const RingView rv = ring_view(*d_total, cap, ring);
    block_select(nblk * k, k,
                 [&](int i, double &v, int64_t &o) {
                     const int b = i / k, e = i - b * k;
                     v = part_s[((size_t)b * Q + q) * k + e];
                     o = part_o[((size_t)b * Q + q) * k + e];
                 },
                 win_s, win_o, red_s, red_o);
    if (tid < k) {
        const int64_t o = win_o[tid];
        double shown = win_s[tid];
        bool ok = o >= 0;
        if (ok) {
            if (score_mode == VM_SCORE_UNIT_INTERVAL) shown = __ddiv_rn(__dadd_rn(1.0, shown), 2.0);
            if (use_min && !(shown > min_score)) ok = false;
        }
        out_scores[(size_t)q * k + tid] = ok ? shown : 0.0;
        out_rows[(size_t)q * k + tid] = ok ? (rv.base + o) * row_stride + row_offset : -1;
    }
}

int redo_blocks(const vm_memory *m) {
    int64_t b = (m->cap + REDO_CHUNK - 1) / REDO_CHUNK;
    if (b > m->ctx->num_cus) b = m->ctx->num_cus;
    return b < 1 ? 1 : (int)b;
}

}  // namespace

extern "C" int vm_cosine_exact(vm_ctx *ctx, const void *queries, int Q, const void *rows, int64_t S, int D,
                               int dtype, double *out, void *stream) {
    if (!ctx || !queries || !out || Q <= 0 || S < 0 || (S > 0 && !rows))
        return vm_fail(ctx, VM_ERR_INVALID, "vm_cosine_exact: bad arguments");
    if (S == 0) return VM_OK;
    dim3 grid((unsigned)((S + 255) / 256), Q);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VM_F32) {
        if (D <= 0) return vm_fail(ctx, VM_ERR_INVALID, "vm_cosine_exact: D=%d", D);
        cosine_exact_f32_kernel<<<grid, 256, 0, st>>>((const float *)queries, (const float *)rows, S, D, out);
        VM_LAUNCH_CHECK(ctx);
        return VM_OK;
    }
    if (D <= 0 || D % 8 != 0) return vm_fail(ctx, VM_ERR_UNSUPPORTED, "vm_cosine_exact: D=%d not a multiple of 8", D);
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
                                          row_offset, out_scores, out_rows, nullptr);
    VM_LAUNCH_CHECK(ctx);
    return VM_OK;
}

extern "C" int vm_topk_select(vm_ctx *ctx, const double *scores, int Q, int64_t S, const int64_t *col_limit, int k,
                              int64_t row_base, double *out_scores, int64_t *out_rows, void *stream) {
    if (!ctx || !scores || !out_scores || !out_rows || Q <= 0 || S < 0 || k <= 0)
        return vm_fail(ctx, VM_ERR_INVALID, "vm_topk_select: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    vm_prof_scope prof(ctx, VM_PROF_TOPK_EXACT, st);
    topk_select_kernel<<<Q, 256, 0, st>>>(scores, S, S, k, 0, 0.0, VM_SCORE_RAW, row_base, 1, 0, out_scores, out_rows,
                                          col_limit);
    VM_LAUNCH_CHECK(ctx);
    return VM_OK;
}

extern "C" size_t vm_topk_redo_workspace_bytes(const vm_memory *m, int Q, int k) {
    if (!m || Q <= 0 || k <= 0 || k > REDO_KMAX) return 0;
    return vm_align_up((size_t)redo_blocks(m) * Q * k * 16, 256);
}

extern "C" int vm_topk_redo_flagged(vm_memory *m, const void *queries, int Q, int k, int use_min_score,
                                    double min_score, int score_mode, int64_t row_stride, int64_t row_offset,
                                    const int32_t *query_flags, double *out_scores, int64_t *out_rows,
                                    void *workspace, size_t workspace_bytes, void *stream) {
    if (!m) return VM_ERR_INVALID;
    vm_ctx *ctx = m->ctx;
    if (!queries || !out_scores || !out_rows || !query_flags || Q <= 0 || k <= 0)
        return vm_fail(ctx, VM_ERR_INVALID, "vm_topk_redo_flagged: bad arguments");
    if (k > REDO_KMAX) return vm_fail(ctx, VM_ERR_UNSUPPORTED, "vm_topk_redo_flagged: k=%d > %d", k, REDO_KMAX);
    if (score_mode != VM_SCORE_RAW && score_mode != VM_SCORE_UNIT_INTERVAL)
        return vm_fail(ctx, VM_ERR_INVALID, "bad score_mode %d", score_mode);
    const int nblk = redo_blocks(m);
    const size_t need = (size_t)nblk * Q * k * 16;
    if (!workspace || workspace_bytes < need)
        return vm_fail(ctx, VM_ERR_NOMEM, "vm_topk_redo_flagged: workspace %zu < %zu", workspace_bytes, need);
    if (((uintptr_t)workspace & 15) || ((uintptr_t)queries & 15))
        return vm_fail(ctx, VM_ERR_INVALID, "vm_topk_redo_flagged: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    vm_prof_scope prof(ctx, VM_PROF_TOPK_EXACT, st);
    double *part_s = (double *)workspace;
    int64_t *part_o = (int64_t *)((char *)workspace + (size_t)nblk * Q * k * 8);
    const size_t lds = (size_t)m->D * 2 + (size_t)REDO_CHUNK * 8;
    if (m->dtype == VM_F16)
        topk_redo_scan_kernel<VM_F16><<<nblk, REDO_THREADS, lds, st>>>(
            (const uint16_t *)queries, m->rows, m->norm64, m->d_total, m->cap, m->ring, m->D, Q, k, query_flags,
            part_s, part_o);
    else
        topk_redo_scan_kernel<VM_BF16><<<nblk, REDO_THREADS, lds, st>>>(
            (const uint16_t *)queries, m->rows, m->norm64, m->d_total, m->cap, m->ring, m->D, Q, k, query_flags,
            part_s, part_o);
    VM_LAUNCH_CHECK(ctx);
    topk_redo_merge_kernel<<<Q, REDO_THREADS, 0, st>>>(part_s, part_o, nblk, Q, k, query_flags, m->d_total, m->cap,
                                                      m->ring, use_min_score, min_score, score_mode, row_stride,
                                                      row_offset, out_scores, out_rows);
    VM_LAUNCH_CHECK(ctx);
    return VM_OK;
}
