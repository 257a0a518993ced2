/* CPU ORACLE (test infrastructure, NOT product code): plain-C restatement of the reference's
 * cosine + stable top-k loop, for sizes the pure-Python restatement cannot finish in seconds.
 *
 *   per pair   : src/components/pre_llm_injector.py:374-388  (three left-to-right fp64 sums, math.sqrt,
 *                zero-norm -> 0.0, dot / (norm1 * norm2))
 *   per query  : src/components/pre_llm_injector.py:356-370  (score every stored row in memory order,
 *                stable descending sort, keep the first k)
 *   threshold  : src/pipeline/retriever_hybrid.py:298        (``similarity > min_score``; optional)
 *
 * Inputs are the SAME quantised values the GPU sees (fp16 / bf16 bit patterns), widened exactly to fp64.
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no FMA contraction, every product and every
 * partial sum is rounded separately, as CPython does).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VMREF_F16 0
#define VMREF_BF16 1
#define VMREF_F32 2

static double half_to_double(uint16_t h) {
    uint32_t sign = (uint32_t)(h >> 15) & 1u;
    uint32_t exp = (uint32_t)(h >> 10) & 0x1Fu;
    uint32_t man = (uint32_t)h & 0x3FFu;
    double v;
    if (exp == 0) {
        v = ldexp((double)man, -24); /* subnormal or zero */
    } else if (exp == 31) {
        v = man ? NAN : INFINITY;
    } else {
        v = ldexp((double)(man | 0x400u), (int)exp - 25);
    }
    return sign ? -v : v;
}

static double bf16_to_double(uint16_t b) {
    uint32_t u = ((uint32_t)b) << 16;
    float f;
    memcpy(&f, &u, 4);
    return (double)f;
}

static void widen_row(const void *src, int dtype, int64_t row, int D, double *dst) {
    if (dtype == VMREF_F32) {
        const float *p = (const float *)src + row * (int64_t)D;
        for (int i = 0; i < D; ++i) dst[i] = (double)p[i];
    } else {
        const uint16_t *p = (const uint16_t *)src + row * (int64_t)D;
        if (dtype == VMREF_F16)
            for (int i = 0; i < D; ++i) dst[i] = half_to_double(p[i]);
        else
            for (int i = 0; i < D; ++i) dst[i] = bf16_to_double(p[i]);
    }
}

/* One reference cosine, sequential sums (pre_llm_injector.py:381-388). */
static double cosine_seq(const double *a, const double *b, int D, double norm_a) {
    double dot = 0.0, nb = 0.0;
    for (int i = 0; i < D; ++i) dot = dot + a[i] * b[i];
    for (int i = 0; i < D; ++i) nb = nb + b[i] * b[i];
    double norm_b = sqrt(nb);
    if (norm_a == 0.0 || norm_b == 0.0) return 0.0;
    return dot / (norm_a * norm_b);
}

/* score_mode: 0 raw cosine, 1 (1+cos)/2.  use_min: 0 none, 1 keep only score > min_score.
 * out_rows[Q*k] (-1 padded), out_scores[Q*k] (0 padded).  Returns 0. */
int vmref_cosine_topk(const void *queries, const void *memory, int dtype, int Q, int64_t M, int D, int k,
                      int score_mode, int use_min, double min_score, int64_t *out_rows, double *out_scores) {
    double *q = (double *)malloc(sizeof(double) * (size_t)D);
    double *m = (double *)malloc(sizeof(double) * (size_t)D);
    if (!q || !m) return -1;
    for (int qi = 0; qi < Q; ++qi) {
        int64_t *rows = out_rows + (int64_t)qi * k;
        double *sc = out_scores + (int64_t)qi * k;
        int n = 0;
        for (int j = 0; j < k; ++j) {
            rows[j] = -1;
            sc[j] = 0.0;
        }
        widen_row(queries, dtype, qi, D, q);
        double nq = 0.0;
        for (int i = 0; i < D; ++i) nq = nq + q[i] * q[i];
        double norm_q = sqrt(nq);
        for (int64_t r = 0; r < M; ++r) {
            widen_row(memory, dtype, r, D, m);
            double s = cosine_seq(q, m, D, norm_q);
            if (score_mode == 1) s = (1.0 + s) / 2.0;
            if (use_min && !(s > min_score)) continue;
            /* stable descending insertion: a later row only passes rows with a strictly smaller score */
            if (n == k && !(s > sc[k - 1])) continue;
            int pos = n < k ? n : k - 1;
            while (pos > 0 && s > sc[pos - 1]) {
                sc[pos] = sc[pos - 1];
                rows[pos] = rows[pos - 1];
                --pos;
            }
            sc[pos] = s;
            rows[pos] = r;
            if (n < k) ++n;
        }
    }
    free(q);
    free(m);
    return 0;
}

/* All-pairs scores [Q,M] fp64 (for the threshold filter / post-compression checks). */
int vmref_cosine_matrix(const void *queries, const void *memory, int dtype, int Q, int64_t M, int D,
                        double *out) {
    double *q = (double *)malloc(sizeof(double) * (size_t)D);
    double *m = (double *)malloc(sizeof(double) * (size_t)D);
    if (!q || !m) return -1;
    for (int qi = 0; qi < Q; ++qi) {
        widen_row(queries, dtype, qi, D, q);
        double nq = 0.0;
        for (int i = 0; i < D; ++i) nq = nq + q[i] * q[i];
        double norm_q = sqrt(nq);
        for (int64_t r = 0; r < M; ++r) {
            widen_row(memory, dtype, r, D, m);
            out[(int64_t)qi * M + r] = cosine_seq(q, m, D, norm_q);
        }
    }
    free(q);
    free(m);
    return 0;
}
