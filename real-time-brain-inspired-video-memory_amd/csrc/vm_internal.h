// Handle layouts private to libvidmem.
#pragma once
#include "vm_common.h"

struct vm_memory {
    vm_ctx *ctx;
    int64_t cap;       // rows
    int D;             // multiple of 128
    int dtype;         // vm_dtype
    int ring;          // 1: overwrite oldest
    uint16_t *rows;    // [cap_pad, D] 16-bit
    double *norm64;    // [cap_pad] exact reference norm of each stored row
    float *rnorm32;    // [cap_pad] 1/norm (0 for a zero row) for the fp32 scan
    int64_t *d_total;  // device: rows appended so far (drives slots under graph replay)
    int64_t h_total;   // host mirror
};

// Logical view of the (ring) row store for a device-side row count: searchable rows n, physical slot of the oldest
// row (head), row id of the oldest row (base).  Row of age order o (0 = oldest) sits in slot (o + head) % cap.
struct RingView {
    int64_t n, head, base, cap;
};
__host__ __device__ inline RingView ring_view(int64_t total, int64_t cap, int ring) {
    RingView v;
    v.cap = cap;
    if (ring && total > cap) {
        v.n = cap;
        v.head = total % cap;
        v.base = total - cap;
    } else {
        v.n = total < cap ? total : cap;
        v.head = 0;
        v.base = 0;
    }
    return v;
}

// The rows whose every score the cut cascade keeps in its first, DENSE pass: the NEWEST stored rows (at most 4,095 of
// them, a whole number of 256-row panels plus the ragged end).  New rows are what a video's current frames resemble
// most - a scene lasts thousands of frames - so the first cut is high and the later passes emit little; with the
// physically first rows as the dense set, a growing (non-ring) memory whose newest few thousand rows all beat the cut of
// the old ones overflowed every query's candidate buffer in the last pass and sent the whole batch to the exhaustive
// redo (correct, 1 s instead of 1 ms: found by the extractor bench leg on a clip processed twice).
// Physical slots [d0, d1); the later passes scan [0, n) in physical order and skip these.
struct DenseRange {
    int64_t d0, d1;
};
__host__ __device__ inline DenseRange dense_newest(const RingView &rv) {
    const int64_t end = rv.head ? rv.head : rv.n;   // physical end (exclusive) of the newest rows
    const int64_t e_al = end & ~(int64_t)255;
    DenseRange r;
    if (e_al >= 3840) {
        r.d0 = e_al - 3840;
        r.d1 = end;
    } else {   // fewer than 15 panels before the end: the physically first rows (they contain the newest ones)
        r.d0 = 0;
        r.d1 = rv.n < 4095 ? rv.n : 4095;
    }
    return r;
}

// ---- emit-only many-query scan (topk_emit.hip), driven by topk.hip ---------------------------------------------
constexpr int VM_EMIT_CAP = 4096;  // candidate slots per query; more -> the query is marked for the exhaustive redo
bool vm_topk_emit_supported(const vm_memory *m, int Q, int KL);
size_t vm_topk_emit_workspace_bytes(int q_pad);
int vm_topk_emit_scan(vm_memory *m, const void *queries, int Q, int q_thr, const float *thr_s, const int *thr_o,
                      int *cand_cnt, float *cand_s, int *cand_o, int64_t row_begin, int64_t row_limit, hipStream_t st);
int vm_topk_emit_compact(vm_memory *m, int Q, int KL, int *cand_cnt, float *cand_s, int *cand_o, float *part_s,
                         int *part_o, int *mark, float *cut_s, int *cut_o, int seed, hipStream_t st);
// ---- GEMM-class scan for very many queries (topk_gscan.hip), reached through vm_topk_emit_scan ---------------------
bool vm_topk_gscan_supported(const vm_memory *m, int Q, int64_t rows);
int vm_topk_gscan(vm_memory *m, const void *queries, int Q, int q_thr, const float *thr_s, const int *thr_o,
                  int *cand_cnt, float *cand_s, int *cand_o, int64_t row_begin, int64_t row_limit, hipStream_t st);
