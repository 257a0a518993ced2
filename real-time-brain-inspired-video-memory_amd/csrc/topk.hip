// Cosine top-k over the resident embedding memory.
//
// Replaces, arithmetic included:
//   src/components/pre_llm_injector.py:346-388  (_calculate_batch_similarities + _cosine_similarity)
//   src/pipeline/retriever_hybrid.py:293-306    (the Cypher cosine scan / ORDER BY / LIMIT)
//
// Two stages.
//   scan     : HBM-bound pass over the [n, D] 16-bit rows.  Each wave takes 16-row tiles; the row tile is the
//              MFMA A operand (loaded straight into registers: every row is read exactly once), the query tile
//              is the B operand (staged once per block in LDS, chunk-swizzled so the ds_read_b128 fragment reads
//              are conflict-free).  fp32 scores * 1/||row||; every lane keeps a sorted list of its KL best
//              (score desc, age-order asc) in registers; lists are merged across the 4 lane groups by shuffles
//              and across waves by rank counting in LDS -> one sorted list of KL per (block, query).
//   finalize : one block per query.  KL-th best list head = threshold; the <= KL*KL entries above it are ranked
//              in LDS; the best KL are re-scored EXACTLY as the reference does (fp64, one rounding per product
//              and per partial sum, left to right; norms likewise; dot / (nq * nm)), ordered by
//              (exact score desc, row id asc) == Python's stable sort over memory order, filtered and written.
//              The result is certified when the exact k-th score clears the best rejected fp32 score by more
//              than the fp32 error bound; otherwise the query is counted in *uncertified and the caller runs
//              the exhaustive kernel (topk_exact.hip).
#include "vm_internal.h"

#include <climits>

namespace {

constexpr int SCAN_THREADS = 512;
constexpr int FIN_THREADS = 256;
constexpr int MAX_BLOCKS = 512;  // lists per query the finalize kernel accepts

__device__ __forceinline__ bool better(float s1, int o1, float s2, int o2) {
    return s1 > s2 || (s1 == s2 && o1 < o2);
}

// Sorted (best first) register list, branch-free insert with compile-time indices only.
template <int KL>
__device__ __forceinline__ void list_insert(float (&ls)[KL], int (&lo)[KL], float s, int o) {
#pragma unroll
    for (int i = KL - 1; i > 0; --i) {
        const bool shift = better(s, o, ls[i - 1], lo[i - 1]);  // new entry lands above i: i takes i-1
        const bool here = better(s, o, ls[i], lo[i]);           // new entry lands at or above i
        ls[i] = shift ? ls[i - 1] : (here ? s : ls[i]);
        lo[i] = shift ? lo[i - 1] : (here ? o : lo[i]);
    }
    const bool here0 = better(s, o, ls[0], lo[0]);
    ls[0] = here0 ? s : ls[0];
    lo[0] = here0 ? o : lo[0];
}

// ---------------------------------------------------------------------------------------------------------
// scan
// ---------------------------------------------------------------------------------------------------------
// LDS: [QT*16][D] query elements (chunk-swizzled); reused afterwards as the merge area [waves][QT*16][KL] {f32,i32}.
template <int DT, int KL, int QT, int LBV = 8>   // LBV: 16-byte row loads per lane and batch (see LB below)
__global__ void __launch_bounds__(SCAN_THREADS)
    topk_scan_kernel(const uint16_t *__restrict__ mem, const float *__restrict__ rnorm,
                     const uint16_t *__restrict__ queries, const int64_t *__restrict__ d_total, int64_t cap,
                     int ring, int D, int Q, int q_pad, float *__restrict__ part_s, int *__restrict__ part_o,
                     int64_t row_limit, const float *__restrict__ thr_s, const int *__restrict__ thr_o, int qgroups,
                     int nt_flag) {
    using E = vm_elem<DT>;
    using vec8 = typename E::vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4 *qlds = reinterpret_cast<uint4 *>(smem);
    const int chunks = D / 8;  // 16-byte chunks per row; multiple of 16
    // the merge area reuses the query area once the scan loop is done (barrier in between)
    float *ms = reinterpret_cast<float *>(smem);
    const int nw = SCAN_THREADS / 64;
    int *mo = reinterpret_cast<int *>(ms + nw * QT * 16 * KL);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, h = lane >> 4;
    // Workgroup -> (row block bx of nbx, query group by).  With several query groups the launch is 1-D and remapped
    // so that the groups of one row block are neighbours ON ONE XCD (workgroup id % 8 picks the XCD): they stream the
    // same rows at the same pace, so one of them pulls a row from HBM / MALL and the others hit that XCD's L2.
    int bx = blockIdx.x, by = blockIdx.y, nbx = gridDim.x;
    if (qgroups > 1) {
        const int total = gridDim.x, id = blockIdx.x;
        int v = id;
        if ((total & 7) == 0) v = (id & 7) * (total >> 3) + (id >> 3);
        nbx = total / qgroups;
        bx = v / qgroups;
        by = v - bx * qgroups;
    }
    const int q0 = by * (QT * 16);

    // stage the query tile: chunk ci of query q sits at (ci & ~15) | ((ci ^ q) & 15)
    for (int idx = tid; idx < QT * 16 * chunks; idx += SCAN_THREADS) {
        const int q = idx / chunks, ci = idx - q * chunks;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (q0 + q < Q) v = reinterpret_cast<const uint4 *>(queries + (size_t)(q0 + q) * D)[ci];
        qlds[q * chunks + ((ci & ~15) | ((ci ^ q) & 15))] = v;
    }
    __syncthreads();

    float ls[QT][KL];
    int lo[QT][KL];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int i = 0; i < KL; ++i) {
            ls[t][i] = -INFINITY;
            lo[t][i] = INT_MAX;
        }

    RingView rv = ring_view(*d_total, cap, ring);
    if (rv.n > row_limit) rv.n = row_limit;  // the sampling pre-pass scans only the first row_limit slots
    const int64_t ntiles = (rv.n + 15) / 16;
    const int ksteps = D / 32;  // multiple of 4
    // Optional per-query cut (score, order) from the sampling pre-pass: the KL-th best of a SUBSET of the rows.  At
    // least KL rows are at least that good, so anything strictly worse cannot be in the top KL and skips the insert.
    float ts[QT];
    int to[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const bool have = thr_s && q0 + 16 * t + r16 < Q;
        ts[t] = have ? thr_s[q0 + 16 * t + r16] : -INFINITY;
        to[t] = have ? thr_o[q0 + 16 * t + r16] : INT_MAX;
    }

    // Row data goes global -> registers in batches of LB 16-byte loads per lane (8 KiB per wave).  The NEXT batch -
    // of this row tile or of the wave's next one - is issued before the MFMAs of the current batch: with two waves per
    // SIMD (the per-lane lists fill the register file) nothing else hides the HBM latency, and without the prefetch
    // a multi-tile scan spent ~9 us per row tile waiting for three round trips.
    constexpr int LB = LBV;
    const int64_t tile_step = (int64_t)nbx * nw;
    const uint4 *qrow = qlds + r16 * chunks;  // this lane's query row of tile 0
    const int tstride = 16 * chunks;          // uint4 units between query tiles
    auto src_of = [&](int64_t tile) {
        int64_t row = tile * 16 + r16;
        if (row > rv.n - 1) row = rv.n - 1;  // tail lanes re-read the last row; their scores are masked below
        return reinterpret_cast<const uint4 *>(mem + (size_t)row * D) + h;
    };
    // one query group = every row byte is read once in this launch: non-temporal loads (VIDMEM_TOPK_NT=0: default
    // policy); with several, the groups of a row block share the rows through their XCD's L2
    typedef unsigned u32x4_nt __attribute__((ext_vector_type(4)));
    const bool nt_rows = nt_flag && qgroups == 1;
    auto issue = [&](const uint4 *src, int s0, uint4 (&a)[LB]) {
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const uint4 *p = src + (s0 + u < ksteps ? s0 + u : ksteps - 1) * 4;
            if (nt_rows) {
                const u32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt *>(p));
                a[u] = make_uint4(v.x, v.y, v.z, v.w);
            } else {
                a[u] = *p;
            }
        }
    };
    // (single-tile scans, QT == 1, run three waves per SIMD and are HBM-bound at 5.5 TB/s without it; there the extra
    // registers and copies cost 15 %, so they keep the plain load-then-use order)
    constexpr bool PREFETCH = QT >= 2 || KL >= 32;  // KL >= 32 runs two waves per SIMD as well
    uint4 cur[LB];
    const int64_t tile0 = (int64_t)bx * nw + wave;
    if (PREFETCH && tile0 < ntiles) issue(src_of(tile0), 0, cur);
    for (int64_t tile = tile0; tile < ntiles; tile += tile_step) {
        const uint4 *src = src_of(tile);
        f32x4 acc[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < ksteps; s0 += LB) {
            uint4 nxt[LB];
            if (!PREFETCH) {
                issue(src, s0, cur);
            } else if (s0 + LB < ksteps) {
                issue(src, s0 + LB, nxt);
            } else if (tile + tile_step < ntiles) {
                issue(src_of(tile + tile_step), 0, nxt);
            }
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                if (s0 + u < ksteps) {  // ksteps is a multiple of 4; uniform branch
                    const int ci = h + 4 * (s0 + u);
                    const vec8 av = __builtin_bit_cast(vec8, cur[u]);
                    // one swizzled address per k-step; the QT query tiles sit a wave-uniform stride apart (spelled
                    // out: left to the compiler, the index arithmetic was ~6 VALU per LDS read, 600 per row tile)
                    const uint4 *qp = qrow + ((ci & ~15) | ((ci ^ r16) & 15));
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        const uint4 bq = qp[t * tstride];
                        acc[t] = E::mfma16(av, __builtin_bit_cast(vec8, bq), acc[t]);
                    }
                }
            }
            if (PREFETCH) {
#pragma unroll
                for (int u = 0; u < LB; ++u) cur[u] = nxt[u];
            }
        }
        // acc[t][j] = <row tile*16 + 4h + j , query q0 + 16t + r16>
        const int64_t p0 = tile * 16 + 4 * h;
        const float4 rn = *reinterpret_cast<const float4 *>(rnorm + p0);  // allocation is padded to 64 rows
        const float rnv[4] = {rn.x, rn.y, rn.z, rn.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t p = p0 + j;
            int64_t o64 = p - rv.head;
            if (o64 < 0) o64 += rv.cap;
            const int o = (int)o64;
            const bool valid = p < rv.n;
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const float s = valid ? acc[t][j] * rnv[j] : -INFINITY;
                if (valid && !better(ts[t], to[t], s, o) && better(s, o, ls[t][KL - 1], lo[t][KL - 1]))
                    list_insert<KL>(ls[t], lo[t], s, o);
            }
        }
    }

    // merge the 4 lane groups of each query inside the wave (partners at lane ^ 16, lane ^ 32): half-cleaner
    // against the partner's reversed list keeps the KL best of both (a bitonic sequence), then a bitonic merge
    // network re-sorts it.  Static indices only, O(KL log KL) code.
#pragma unroll
    for (int step = 16; step <= 32; step <<= 1) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float ps[KL];
            int po[KL];
#pragma unroll
            for (int i = 0; i < KL; ++i) {
                ps[i] = __shfl_xor(ls[t][KL - 1 - i], step, 64);
                po[i] = __shfl_xor(lo[t][KL - 1 - i], step, 64);
            }
#pragma unroll
            for (int i = 0; i < KL; ++i) {
                const bool take = better(ps[i], po[i], ls[t][i], lo[t][i]);
                ls[t][i] = take ? ps[i] : ls[t][i];
                lo[t][i] = take ? po[i] : lo[t][i];
            }
#pragma unroll
            for (int stride = KL / 2; stride > 0; stride >>= 1) {
#pragma unroll
                for (int i = 0; i < KL; ++i) {
                    if ((i & stride) == 0) {
                        const bool sw = better(ls[t][i + stride], lo[t][i + stride], ls[t][i], lo[t][i]);
                        const float s_hi = sw ? ls[t][i + stride] : ls[t][i];
                        const float s_lo = sw ? ls[t][i] : ls[t][i + stride];
                        const int o_hi = sw ? lo[t][i + stride] : lo[t][i];
                        const int o_lo = sw ? lo[t][i] : lo[t][i + stride];
                        ls[t][i] = s_hi;
                        ls[t][i + stride] = s_lo;
                        lo[t][i] = o_hi;
                        lo[t][i + stride] = o_lo;
                    }
                }
            }
        }
    }
    __syncthreads();  // every wave is done reading the query tile: its LDS becomes the merge area
    if (h == 0) {
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int i = 0; i < KL; ++i) {
                ms[((wave * QT + t) * 16 + r16) * KL + i] = ls[t][i];
                mo[((wave * QT + t) * 16 + r16) * KL + i] = lo[t][i];
            }
    }
    __syncthreads();
    // cross-wave merge with the same network: query tile t goes to wave t; its lane (r16, h) picks up the lists of
    // waves h, h + 4, ... for query (t, r16), merges them in registers, and the four lane groups merge as above.
    // (The first version ranked all nw*KL candidates against each other out of LDS: ~10 k VALU instructions per wave,
    // as much as scanning 30 row tiles, and the reason more row-blocks made multi-tile scans slower.)
    static_assert(SCAN_THREADS % 256 == 0, "the cross-wave merge maps waves 4g + h onto lane group h");
    for (int t = wave; t < QT; t += nw) {
        float fs[KL];
        int fo[KL];
#pragma unroll
        for (int i = 0; i < KL; ++i) {
            fs[i] = ms[((h * QT + t) * 16 + r16) * KL + i];
            fo[i] = mo[((h * QT + t) * 16 + r16) * KL + i];
        }
#pragma unroll 1  // ONE copy of the network in the code: unrolled, this block took the compiler > 30 minutes
        for (int round = 1; round < nw / 4 + 2; ++round) {
            // rounds 1 .. nw/4-1: list of wave h + 4*round from LDS; last two rounds: lane partners ^16, ^32
            float ps[KL];
            int po[KL];
            if (round < nw / 4) {
#pragma unroll
                for (int i = 0; i < KL; ++i) {
                    ps[i] = ms[(((h + 4 * round) * QT + t) * 16 + r16) * KL + (KL - 1 - i)];
                    po[i] = mo[(((h + 4 * round) * QT + t) * 16 + r16) * KL + (KL - 1 - i)];
                }
            } else {
                const int step = round == nw / 4 ? 16 : 32;
#pragma unroll
                for (int i = 0; i < KL; ++i) {
                    ps[i] = __shfl_xor(fs[KL - 1 - i], step, 64);
                    po[i] = __shfl_xor(fo[KL - 1 - i], step, 64);
                }
            }
#pragma unroll
            for (int i = 0; i < KL; ++i) {
                const bool take = better(ps[i], po[i], fs[i], fo[i]);
                fs[i] = take ? ps[i] : fs[i];
                fo[i] = take ? po[i] : fo[i];
            }
#pragma unroll
            for (int stride = KL / 2; stride > 0; stride >>= 1) {
#pragma unroll
                for (int i = 0; i < KL; ++i) {
                    if ((i & stride) == 0) {
                        const bool sw = better(fs[i + stride], fo[i + stride], fs[i], fo[i]);
                        const float s_hi = sw ? fs[i + stride] : fs[i];
                        const float s_lo = sw ? fs[i] : fs[i + stride];
                        const int o_hi = sw ? fo[i + stride] : fo[i];
                        const int o_lo = sw ? fo[i] : fo[i + stride];
                        fs[i] = s_hi;
                        fs[i + stride] = s_lo;
                        fo[i] = o_hi;
                        fo[i + stride] = o_lo;
                    }
                }
            }
        }
        if (h == 0) {
            const size_t dst = ((size_t)bx * q_pad + q0 + t * 16 + r16) * KL;
#pragma unroll
            for (int i = 0; i < KL; ++i) {
                part_s[dst + i] = fs[i];
                part_o[dst + i] = fo[i];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// finalize
// ---------------------------------------------------------------------------------------------------------
// THRESH: stop after the fp32 ranking and publish the KL-th best (score, order) per query for the main scan
template <int DT, int KL, bool THRESH>
__global__ void __launch_bounds__(FIN_THREADS)
    topk_finalize_kernel(const uint16_t *__restrict__ mem, const double *__restrict__ norm64,
                         const uint16_t *__restrict__ queries, const int64_t *__restrict__ d_total, int64_t cap,
                         int ring, int D, int q_pad, int nblk, const float *__restrict__ part_s,
                         const int *__restrict__ part_o, int k, int use_min, double min_score, int score_mode,
                         int64_t row_stride, int64_t row_offset, double *__restrict__ out_scores,
                         int64_t *__restrict__ out_rows, int *__restrict__ uncertified,
                         int *__restrict__ qflags, float *__restrict__ thr_s_out, int *__restrict__ thr_o_out, int stage_rows,
                         const int *__restrict__ mark) {
    using E = vm_elem<DT>;
    __shared__ float hs[MAX_BLOCKS];
    __shared__ int ho[MAX_BLOCKS];
    __shared__ int qual[KL];
    __shared__ float cs[KL * KL];
    __shared__ int co[KL * KL];
    __shared__ float fs[KL];
    __shared__ int fo[KL];
    __shared__ double ex[KL];
    __shared__ double qnorm_sh;
    __shared__ float qsq_sh[FIN_THREADS / 64];
    __shared__ int cnt, nqual;
    __shared__ float cut_s_sh;
    __shared__ int cut_o_sh;
    extern __shared__ __attribute__((aligned(16))) char fin_dyn[];  // the query row, [D] 16-bit
    uint16_t *ql = reinterpret_cast<uint16_t *>(fin_dyn);

    const int q = blockIdx.x, tid = threadIdx.x;
    const RingView rv = ring_view(*d_total, cap, ring);

    if (tid == 0) {
        cnt = 0;
        nqual = 0;
        if (!THRESH && qflags) qflags[q] = 0;  // set again below by the thread that fails to certify this query
    }
    {   // stage the query row in LDS; fp32 upper bound of |q| (gates the any-order-exact fast paths below)
        float sq = 0.f;
        for (int i = tid; i < D / 8; i += FIN_THREADS) {
            const uint4 v = reinterpret_cast<const uint4 *>(queries + (size_t)q * D)[i];
            reinterpret_cast<uint4 *>(ql)[i] = v;
            const uint16_t *e = reinterpret_cast<const uint16_t *>(&v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = E::to_float(e[j]);
                sq += x * x;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
        if ((tid & 63) == 0) qsq_sh[tid >> 6] = sq;
    }
    for (int b = tid; b < nblk; b += FIN_THREADS) {
        hs[b] = part_s[((size_t)b * q_pad + q) * KL];
        ho[b] = part_o[((size_t)b * q_pad + q) * KL];
    }
    if (tid < KL) {
        fs[tid] = -INFINITY;
        fo[tid] = INT_MAX;
    }
    __syncthreads();
    // Lists whose head ranks among the KL best heads (score desc, order asc) can contain members of the global top
    // KL.  One wave finds that cut with two bitwise binary searches held entirely in registers / SGPR ballots (the
    // KL-th largest score key, then among equal scores the needed count of smallest orders): ~1.3k instructions
    // instead of an O(lists^2) LDS rank count.
    if (nblk == 1) {
        // one list (the emit / GEMM-class scans hand over a single compacted list per query): it qualifies as it is - no
        // head search (31 ballot rounds over MAX_BLOCKS list heads), no cut
        if (tid == 0) {
            const bool have = hs[0] > -INFINITY;
            qual[0] = 0;
            nqual = have ? 1 : 0;
            cut_s_sh = -INFINITY;
            cut_o_sh = INT_MAX;
        }
    } else if (tid < 64) {
        constexpr int HPL = MAX_BLOCKS / 64;
        unsigned key[HPL];
        int ord[HPL];
#pragma unroll
        for (int i = 0; i < HPL; ++i) {
            const int b = tid + 64 * i;
            const bool have = b < nblk && hs[b] > -INFINITY;
            const unsigned u = __builtin_bit_cast(unsigned, have ? hs[b] : 0.f);
            key[i] = have ? ((u & 0x80000000u) ? ~u : (u | 0x80000000u)) : 0u;  // order-preserving; 0 = no list
            ord[i] = have ? ho[b] : INT_MAX;
        }
        auto count_ge = [&](unsigned cand) {
            int c = 0;
#pragma unroll
            for (int i = 0; i < HPL; ++i) c += __popcll(__ballot(key[i] >= cand));
            return c;
        };
        const int total = count_ge(1u);
        const int want = total < KL ? total : KL;
        unsigned tk = 0;  // largest key with count(key >= tk) >= want
        if (want > 0) {
            for (int bit = 31; bit >= 0; --bit) {
                const unsigned cand = tk | (1u << bit);
                if (count_ge(cand) >= want) tk = cand;
            }
        }
        int n_gt = 0;
#pragma unroll
        for (int i = 0; i < HPL; ++i) n_gt += __popcll(__ballot(key[i] > tk));
        const int need_eq = want - n_gt;  // >= 1 when want > 0
        int to = 0;                        // smallest order with count(key == tk && ord <= to) >= need_eq
        if (want > 0) {
            auto count_eq_le = [&](int lim) {
                int c = 0;
#pragma unroll
                for (int i = 0; i < HPL; ++i) c += __popcll(__ballot(key[i] == tk && ord[i] <= lim));
                return c;
            };
            int lo = 0, hi = INT_MAX - 1;  // orders are distinct non-negative ints
            while (lo < hi) {
                const int mid = lo + (hi - lo) / 2;
                if (count_eq_le(mid) >= need_eq) hi = mid; else lo = mid + 1;
            }
            to = lo;
        }
#pragma unroll
        for (int i = 0; i < HPL; ++i) {
            const bool take = want > 0 && key[i] != 0u && (key[i] > tk || (key[i] == tk && ord[i] <= to));
            if (take) qual[atomicAdd(&nqual, 1)] = tid + 64 * i;
        }
        if (tid == 0) {
            // The KL-th best HEAD is a lower bound of the KL-th best candidate overall (KL distinct candidates are at
            // least that good), so members of the qualified lists strictly below it cannot reach the top KL: they
            // are dropped before the O(C^2) rank count (C <= KL^2 without the cut: 0.4 ms per launch at KL = 32).
            const bool have_cut = want == KL;
            const unsigned u = (tk & 0x80000000u) ? (tk & 0x7fffffffu) : ~tk;
            cut_s_sh = have_cut ? __builtin_bit_cast(float, u) : -INFINITY;
            cut_o_sh = have_cut ? to : INT_MAX;
        }
    }
    __syncthreads();
    const int nq_lists = nqual;
    const float cut_s = cut_s_sh;
    const int cut_o = cut_o_sh;
    for (int p = tid; p < nq_lists * KL; p += FIN_THREADS) {
        const int b = qual[p / KL], e = p % KL;
        const float s = part_s[((size_t)b * q_pad + q) * KL + e];
        const int o = part_o[((size_t)b * q_pad + q) * KL + e];
        if (s > -INFINITY && !better(cut_s, cut_o, s, o)) {
            const int slot = atomicAdd(&cnt, 1);
            cs[slot] = s;
            co[slot] = o;
        }
    }
    __syncthreads();
    const int C = cnt;
    for (int c = tid; c < C; c += FIN_THREADS) {
        const float s = cs[c];
        const int o = co[c];
        int rank = 0;
        for (int d = 0; d < C; ++d) rank += better(cs[d], co[d], s, o) ? 1 : 0;
        if (rank < KL) {
            fs[rank] = s;
            fo[rank] = o;
        }
    }
    __syncthreads();
    const int nfin = C < KL ? C : KL;
    if (THRESH) {
        if (tid == 0) {
            thr_s_out[q] = C >= KL ? fs[KL - 1] : -INFINITY;
            thr_o_out[q] = C >= KL ? fo[KL - 1] : INT_MAX;
        }
        return;
    }

    // exact re-scoring.  TPC = 256/KL threads share one candidate.
    //  * fp16 rows with |q||m| < 32: fp16 x fp16 products are exact multiples of 2^-48 and every partial sum stays
    //    below 32, so every partial sum is exactly representable in fp64: NO rounding occurs in ANY summation order,
    //    and a parallel tree gives the reference's left-to-right result bit for bit (sum|q_i m_i| <= |q||m| bounds all
    //    partial sums).  The TPC threads take interleaved 16-byte chunks and combine with shuffles.
    //  * otherwise (bf16, or large norms): one thread sums strictly left to right, as the reference does.
    // The query norm is handled the same way by the last wave's first TPC threads' neighbours (slot KL).
    const uint16_t *qv = ql;  // LDS copy of the query row
    // bf16 has no order-free fast path (its products span too many binades), so every candidate is summed strictly
    // left to right by one thread.  Fed from global memory that chain waits one load round trip per 32 elements
    // (0.43 ms per launch at k = 20, D = 1024); the candidates' rows are therefore staged in LDS first, by all
    // threads, coalesced.  Row pitch D + 8 elements: the one-thread-per-row readers land on different banks.
    const bool STAGED = DT == VM_BF16 && stage_rows;  // block-uniform
    const int RS = D + 8;
    uint16_t *rows_l = ql + D;
    if (STAGED) {
        const int cpr = D / 8;
        for (int idx = tid; idx < nfin * cpr; idx += FIN_THREADS) {
            const int c = idx / cpr, ch = idx - c * cpr;
            int64_t p = fo[c] + rv.head;
            if (p >= rv.cap) p -= rv.cap;
            *reinterpret_cast<uint4 *>(rows_l + (size_t)c * RS + ch * 8) =
                *reinterpret_cast<const uint4 *>(mem + (size_t)p * D + ch * 8);
        }
        __syncthreads();
    }
    const double qnorm_fast = 1.001 * sqrt((double)((qsq_sh[0] + qsq_sh[1]) + (qsq_sh[2] + qsq_sh[3])));
    constexpr int TPC = FIN_THREADS / KL;  // 4 .. 32, a power of two dividing the wave
    {
        const int c = tid / TPC, sub = tid % TPC;
        const bool live = c < nfin;
        int64_t p = live ? fo[c] + rv.head : 0;
        if (p >= rv.cap) p -= rv.cap;
        const uint16_t *mv = STAGED ? rows_l + (size_t)(live ? c : 0) * RS : mem + (size_t)p * D;
        const bool any_order_exact = (DT == VM_F16) && live && (qnorm_fast * norm64[p] < 32.0);
        double dot = 0.0;
        if (any_order_exact) {
            double d0 = 0.0, d1 = 0.0;
            for (int ch = sub; ch < D / 8; ch += TPC) {
                const uint4 a = *reinterpret_cast<const uint4 *>(qv + ch * 8);
                const uint4 b = *reinterpret_cast<const uint4 *>(mv + ch * 8);
                const uint16_t *ae = reinterpret_cast<const uint16_t *>(&a);
                const uint16_t *be = reinterpret_cast<const uint16_t *>(&b);
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    d0 = __dadd_rn(d0, __dmul_rn(E::to_double(ae[j]), E::to_double(be[j])));
                    d1 = __dadd_rn(d1, __dmul_rn(E::to_double(ae[j + 1]), E::to_double(be[j + 1])));
                }
            }
            dot = __dadd_rn(d0, d1);
        } else if (live && sub == 0) {
            for (int i = 0; i < D; i += 32) {  // D is a multiple of 128; 4 row chunks in flight per step
                uint4 b4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) b4[u] = *reinterpret_cast<const uint4 *>(mv + i + 8 * u);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint4 a = *reinterpret_cast<const uint4 *>(qv + i + 8 * u);
                    const uint16_t *ae = reinterpret_cast<const uint16_t *>(&a);
                    const uint16_t *be = reinterpret_cast<const uint16_t *>(&b4[u]);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        dot = __dadd_rn(dot, __dmul_rn(E::to_double(ae[j]), E::to_double(be[j])));
                }
            }
        }
#pragma unroll
        for (int off = TPC / 2; off > 0; off >>= 1) dot = __dadd_rn(dot, __shfl_xor(dot, off, 64));  // zeros elsewhere
        if (live && sub == 0) ex[c] = dot;
    }
    if (tid >= FIN_THREADS - 64) {  // last wave: the query norm (src/components/pre_llm_injector.py:382)
        const int l = tid - (FIN_THREADS - 64);
        const bool q_any_order = (DT == VM_F16) && (qnorm_fast * qnorm_fast < 32.0);  // wave-uniform
        double nq = 0.0;
        if (q_any_order) {  // squares of fp16 values: exact, and every partial sum < 32 -> order-free
            for (int ch = l; ch < D / 8; ch += 64) {
                const uint4 a = *reinterpret_cast<const uint4 *>(qv + ch * 8);
                const uint16_t *ae = reinterpret_cast<const uint16_t *>(&a);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double x = E::to_double(ae[j]);
                    nq = __dadd_rn(nq, __dmul_rn(x, x));
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) nq = __dadd_rn(nq, __shfl_xor(nq, off, 64));
        } else if (l == 0) {  // strictly left to right
            for (int i = 0; i < D; i += 8) {
                const uint4 a = *reinterpret_cast<const uint4 *>(qv + i);
                const uint16_t *ae = reinterpret_cast<const uint16_t *>(&a);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double x = E::to_double(ae[j]);
                    nq = __dadd_rn(nq, __dmul_rn(x, x));
                }
            }
        }
        if (l == 0) qnorm_sh = __dsqrt_rn(nq);
    }
    __syncthreads();
    const double qn = qnorm_sh;
    if (tid < nfin) {
        int64_t p = fo[tid] + rv.head;
        if (p >= rv.cap) p -= rv.cap;
        const double mn = norm64[p];
        // src/components/pre_llm_injector.py:385-388
        ex[tid] = (qn == 0.0 || mn == 0.0) ? 0.0 : __ddiv_rn(ex[tid], __dmul_rn(qn, mn));
    }
    if (tid < k) {
        out_scores[(size_t)q * k + tid] = 0.0;
        out_rows[(size_t)q * k + tid] = -1;
    }
    __syncthreads();
    if (tid < nfin) {
        const double e = ex[tid];
        const int o = fo[tid];
        int rank = 0;
        for (int d = 0; d < nfin; ++d) rank += (ex[d] > e || (ex[d] == e && fo[d] < o)) ? 1 : 0;
        const double shown = score_mode == VM_SCORE_UNIT_INTERVAL ? __ddiv_rn(__dadd_rn(1.0, e), 2.0) : e;
        const bool pass = !use_min || shown > min_score;
        if (rank < k && pass) {
            out_scores[(size_t)q * k + rank] = shown;
            out_rows[(size_t)q * k + rank] = (rv.base + o) * row_stride + row_offset;
        }
        // certification: is the exact k-th score provably above every row that never became a candidate?
        const int kth = (k < nfin ? k : nfin) - 1;
        if (rank == kth && (uncertified || qflags)) {
            const bool all_rows_are_candidates = rv.n <= (int64_t)nfin;
            if (mark && mark[q]) {  // emit scan: candidate buffer overflowed / unranked ties -> exhaustive redo
                if (uncertified) atomicAdd(uncertified, 1);
                if (qflags) qflags[q] = VM_FLAG_OVERFLOW;
            } else if (!all_rows_are_candidates && qn != 0.0) {
                const float bound_f32 = fs[KL - 1];  // best possible fp32 score of a rejected row (x 1/||q||)
                const double eps = 2.0 * (double)(D + 8) * 5.9604644775390625e-08;  // 2*(D+8)*2^-24
                const double reject = (double)bound_f32 / qn + eps;
                if (!(e > reject)) {
                    if (uncertified) atomicAdd(uncertified, 1);
                    if (qflags) qflags[q] = VM_FLAG_GAP;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// merge of per-shard results (after the RCCL all-gather): parts*k entries per query, rank counting
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    topk_merge_kernel(const double *__restrict__ scores, const int64_t *__restrict__ rows, int parts, int Q,
                      int k, double *__restrict__ out_scores, int64_t *__restrict__ out_rows) {
    const int q = blockIdx.x;
    const int n = parts * k;
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        out_scores[(size_t)q * k + i] = 0.0;
        out_rows[(size_t)q * k + i] = -1;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
        const int cp = c / k, ci = c - cp * k;
        const double s = scores[((size_t)cp * Q + q) * k + ci];
        const int64_t r = rows[((size_t)cp * Q + q) * k + ci];
        if (r < 0) continue;
        int rank = 0;
        for (int d = 0; d < n; ++d) {
            const int dp = d / k, di = d - dp * k;
            const double s2 = scores[((size_t)dp * Q + q) * k + di];
            const int64_t r2 = rows[((size_t)dp * Q + q) * k + di];
            if (r2 < 0) continue;
            rank += (s2 > s || (s2 == s && (r2 < r || (r2 == r && d < c)))) ? 1 : 0;
        }
        if (rank < k) {
            out_scores[(size_t)q * k + rank] = s;
            out_rows[(size_t)q * k + rank] = r;
        }
    }
}

constexpr int64_t SAMPLE_ROWS = 16384;  // rows of the sampling pre-pass (multi-tile query groups only)

struct ScanCfg {
    int KL, QT;
};
ScanCfg pick_cfg(int Q, int k, int D) {
    ScanCfg c;
    c.KL = k + 2 <= 8 ? 8 : (k + 4 <= 16 ? 16 : (k + 6 <= 32 ? 32 : 64));
    int qt_max = 64 / c.KL;  // QT*KL <= 64 list registers pairs per lane
    if (qt_max > 4) qt_max = 4;
    int need = (Q + 15) / 16;
    c.QT = need >= 4 && qt_max >= 4 ? 4 : (need >= 2 && qt_max >= 2 ? 2 : 1);
    while (c.QT > 1 && (size_t)c.QT * 16 * D * 2 > 144 * 1024) c.QT /= 2;  // query tile must fit in LDS
    return c;
}

struct ScanPlan {
    ScanCfg cfg;
    int q_pad, qgroups, nblk;
    size_t lds;
    size_t part_bytes;
    bool emit;  // many queries: query-stationary emit scan (topk_emit.hip) instead of the per-lane-list scan
};
ScanPlan make_plan(const vm_memory *m, int Q, int k) {
    ScanPlan p;
    p.cfg = pick_cfg(Q, k, m->D);
    const int qpg = p.cfg.QT * 16;
    p.qgroups = (Q + qpg - 1) / qpg;
    p.q_pad = p.qgroups * qpg;
    const int nw = SCAN_THREADS / 64;
    const int64_t ntiles = (m->cap + 15) / 16;
    int64_t want = (ntiles + nw - 1) / nw;
    static const int env_per_cu = (int)VM_DEV_ENV("TOPK_BLOCKS_PER_CU", 0);
    const int per_cu = env_per_cu > 0 ? env_per_cu : ((p.cfg.KL <= 16 && p.cfg.QT * p.cfg.KL <= 32) ? 2 : 1);
    // Row-blocks per query group.  One group (Q <= 64): as many as the chip holds, the scan is HBM-bound.  Many
    // groups: the groups already fill the chip, and FEWER row-blocks per group means more rows per lane list, so
    // the lists warm up and most scores fail the one-compare threshold test instead of paying a sorted insert.
    int64_t lim = (int64_t)m->ctx->num_cus * per_cu / p.qgroups;
    if (lim < 8) lim = 8;
    if (lim > MAX_BLOCKS) lim = MAX_BLOCKS;
    p.nblk = (int)(want < lim ? want : lim);
    if (p.nblk < 1) p.nblk = 1;
    const size_t lds_q = (size_t)qpg * m->D * 2, lds_m = (size_t)nw * qpg * p.cfg.KL * 8;
    p.lds = lds_q > lds_m ? lds_q : lds_m;
    p.part_bytes = vm_align_up((size_t)p.nblk * p.q_pad * p.cfg.KL * 4, 256);
    p.emit = vm_topk_emit_supported(m, Q, p.cfg.KL) && m->cap >= 4 * SAMPLE_ROWS;
    return p;
}


template <int DT, int KL, int QT>
int launch_scan(vm_memory *m, const ScanPlan &p, int nblk, int64_t row_limit, const float *thr_s, const int *thr_o,
                const void *queries, int Q, float *part_s, int *part_o, hipStream_t st) {
    auto kern = topk_scan_kernel<DT, KL, QT>;
#ifdef VM_DEV_SWITCHES   // developer A/B (VIDMEM_SCAN_LB = 12 | 24): row loads per batch of the one-tile f16 scan
    if constexpr (QT == 1 && KL == 16 && DT == VM_F16) {
        static const int lb_env = (int)VM_DEV_ENV("SCAN_LB", 8);
        if (lb_env == 12) kern = topk_scan_kernel<DT, KL, QT, 12>;
        if (lb_env == 24) kern = topk_scan_kernel<DT, KL, QT, 24>;
    }
#endif
    if (p.lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)p.lds);
        if (e != hipSuccess) return vm_fail(m->ctx, VM_ERR_HIP, "LDS opt-in %zu: %s", p.lds, hipGetErrorString(e));
    }
    const dim3 grid = p.qgroups > 1 ? dim3(nblk * p.qgroups) : dim3(nblk);
    static const int nt_env = (int)VM_DEV_ENV("TOPK_NT", 1);
    const int nt_flag = nt_env && row_limit >= m->cap;  // full passes only: the sampling pre-pass's rows are read again
    vm_prof_scope prof(m->ctx, VM_PROF_TOPK_SCAN, st);
    kern<<<grid, SCAN_THREADS, p.lds, st>>>(m->rows, m->rnorm32, (const uint16_t *)queries, m->d_total, m->cap,
                                          m->ring, m->D, Q, p.q_pad, part_s, part_o, row_limit, thr_s, thr_o, p.qgroups,
                                          nt_flag);
    VM_LAUNCH_CHECK(m->ctx);
    return VM_OK;
}

template <int DT, int KL>
int launch_scan_qt(vm_memory *m, const ScanPlan &p, int nblk, int64_t row_limit, const float *thr_s,
                   const int *thr_o, const void *queries, int Q, float *part_s, int *part_o, hipStream_t st) {
    if constexpr (KL <= 16) {
        if (p.cfg.QT == 4)
            return launch_scan<DT, KL, 4>(m, p, nblk, row_limit, thr_s, thr_o, queries, Q, part_s, part_o, st);
    }
    if constexpr (KL <= 32) {
        if (p.cfg.QT == 2)
            return launch_scan<DT, KL, 2>(m, p, nblk, row_limit, thr_s, thr_o, queries, Q, part_s, part_o, st);
    }
    return launch_scan<DT, KL, 1>(m, p, nblk, row_limit, thr_s, thr_o, queries, Q, part_s, part_o, st);
}

template <int DT, int KL>
int run_topk_kl(vm_memory *m, const ScanPlan &p, const void *queries, int Q, int k, int use_min, double min_score,
                int score_mode, int64_t row_stride, int64_t row_offset, double *out_scores, int64_t *out_rows,
                int *uncertified, int *qflags, float *part_s, int *part_o, float *thr_s, int *thr_o, hipStream_t st) {
    int rc;
    const float *use_ts = nullptr;
    const int *use_to = nullptr;
    const int *mark = nullptr;
    int fin_nblk = p.nblk;
    // Multi-tile query groups are insert-bound, not HBM-bound: a cheap pre-pass over the first sample rows gives
    // every query a valid cut (the KL-th best of that subset); the full scan then either skips the sorted insert for
    // everything below it (list scan) or emits only what is at or above it (emit scan, topk_emit.hip).
    const bool emit = p.emit;
    if (emit) {
        // Cut cascade (topk_emit.hip / topk_gscan.hip), INCREMENTAL: pass 0 keeps every score of the NEWEST ~4 k
        // rows; pass p scans only the slots [limit[p-1], limit[p]) against the KL-th best of the rows seen so far
        // and appends to the KL survivors the compact kernel seeded the buffer with.  Each cut is the KL-th best of a
        // SUBSET of the rows, so the KL best of that subset plus everything at or above the cut among the other rows
        // contain the KL best overall; no row is scanned twice.  With the limits growing 8-fold a pass emits about
        // 7 KL candidates per query whatever the memory size (the previous three-pass cascade re-scanned the sample
        // and emitted rows * KL / sample ~ 1 k per query in its last pass: at 7,040 queries a quarter of all 16 x 16
        // score blocks then took the emission path).
        int *cand_cnt = (int *)((char *)thr_s + vm_align_up((size_t)p.q_pad * 8, 256));
        int *mk = (int *)((char *)cand_cnt + vm_align_up((size_t)p.q_pad * 4, 256));
        float *cand_s = (float *)((char *)mk + vm_align_up((size_t)p.q_pad * 4, 256));
        int *cand_o = (int *)((char *)cand_s + vm_align_up((size_t)p.q_pad * VM_EMIT_CAP * 4, 256));
        // growth of the pass limits: 8 for many queries (a pass emits ~ln(growth) KL candidates per query and every
        // candidate is matrix-pipe time there); 32 when one superblock of <= 128 queries scans at the HBM rate, where a
        // pass costs its two launches (scan + compact, ~25 us) and nothing else: 3 passes instead of 4 over 1 M rows
        static const int growth_many = VM_DEV_ENV("CUT_GROWTH", 8) < 2 ? 2 : (int)VM_DEV_ENV("CUT_GROWTH", 8);
        static const int growth_few = VM_DEV_ENV("CUT_GROWTH_FEW", 32) < 2 ? 2 : (int)VM_DEV_ENV("CUT_GROWTH_FEW", 32);
        const int growth = Q <= 128 ? growth_few : growth_many;
        // cand_cnt and mk are neighbours: one memset clears both
        hipError_t e = hipMemsetAsync(cand_cnt, 0, 2 * vm_align_up((size_t)p.q_pad * 4, 256), st);
        if (e != hipSuccess) return vm_fail(m->ctx, VM_ERR_HIP, "memset: %s", hipGetErrorString(e));
        // pass 0: DENSE over the newest <= 4,095 rows (dense_newest: their physical range is computed on the device);
        // passes 1.. : the physical slots [0, 32768), [32768, 262144), ... minus the dense rows
        int64_t begin = 0, limit = 0;
        for (int pass = 0;; ++pass) {
            const bool last = pass > 0 && limit >= m->cap;
            if ((rc = vm_topk_emit_scan(m, queries, Q, p.q_pad, pass ? thr_s : nullptr, pass ? thr_o : nullptr, cand_cnt,
                                        cand_s, cand_o, begin, last ? INT64_MAX : limit, st)) != VM_OK)
                return rc;
            if ((rc = vm_topk_emit_compact(m, Q, KL, cand_cnt, cand_s, cand_o, part_s, part_o, mk, last ? nullptr : thr_s,
                                           last ? nullptr : thr_o, last ? 0 : 1, st)) != VM_OK)
                return rc;
            if (last) break;
            begin = limit;
            limit = pass == 0 ? (int64_t)VM_EMIT_CAP * growth : limit * growth;
        }
        mark = mk;
        fin_nblk = 1;
    } else {
        if (p.cfg.QT >= 2 && m->cap >= 4 * SAMPLE_ROWS) {  // 100k-row shard, 880 queries: scan 1.06 -> 0.93 ms
            const int nw = SCAN_THREADS / 64;
            int nblk_pre = (int)((SAMPLE_ROWS / 16 + nw - 1) / nw);
            if (nblk_pre > p.nblk) nblk_pre = p.nblk;
            if ((rc = launch_scan_qt<DT, KL>(m, p, nblk_pre, SAMPLE_ROWS, nullptr, nullptr, queries, Q, part_s, part_o,
                                             st)) != VM_OK)
                return rc;
            {
                vm_prof_scope prof(m->ctx, VM_PROF_TOPK_FINALIZE, st);
                topk_finalize_kernel<DT, KL, true><<<Q, FIN_THREADS, (size_t)m->D * 2, st>>>(
                    m->rows, m->norm64, (const uint16_t *)queries, m->d_total, m->cap, m->ring, m->D, p.q_pad, nblk_pre,
                    part_s, part_o, k, use_min, min_score, score_mode, row_stride, row_offset, out_scores, out_rows,
                    nullptr, nullptr, thr_s, thr_o, 0, nullptr);
                VM_LAUNCH_CHECK(m->ctx);
            }
            use_ts = thr_s;
            use_to = thr_o;
        }
        if ((rc = launch_scan_qt<DT, KL>(m, p, p.nblk, INT64_MAX, use_ts, use_to, queries, Q, part_s, part_o, st)) !=
            VM_OK)
            return rc;
    }
    vm_prof_scope prof(m->ctx, VM_PROF_TOPK_FINALIZE, st);
    size_t fin_lds = (size_t)m->D * 2;
    int stage_rows = 0;
    if (DT == VM_BF16) {  // + the KL candidate rows when they fit beside the kernel's static tables (csrc: STAGED)
        const size_t with_rows = fin_lds + (size_t)KL * (m->D + 8) * 2;
        if (with_rows <= 96 * 1024) {
            fin_lds = with_rows;
            stage_rows = 1;
            if (fin_lds > 48 * 1024) {
                hipError_t e = hipFuncSetAttribute((const void *)topk_finalize_kernel<DT, KL, false>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)fin_lds);
                if (e != hipSuccess)
                    return vm_fail(m->ctx, VM_ERR_HIP, "LDS opt-in %zu: %s", fin_lds, hipGetErrorString(e));
            }
        }
    }
    topk_finalize_kernel<DT, KL, false><<<Q, FIN_THREADS, fin_lds, st>>>(
        m->rows, m->norm64, (const uint16_t *)queries, m->d_total, m->cap, m->ring, m->D, p.q_pad, fin_nblk, part_s,
        part_o, k, use_min, min_score, score_mode, row_stride, row_offset, out_scores, out_rows, uncertified, qflags,
        nullptr, nullptr, stage_rows, mark);
    VM_LAUNCH_CHECK(m->ctx);
    return VM_OK;
}

template <int DT>
int run_topk(vm_memory *m, const ScanPlan &p, const void *queries, int Q, int k, int use_min, double min_score,
             int score_mode, int64_t row_stride, int64_t row_offset, double *out_scores, int64_t *out_rows,
             int *uncertified, int *qflags, float *part_s, int *part_o, float *thr_s, int *thr_o, hipStream_t st) {
#define GO(KLV)                                                                                                   \
    return run_topk_kl<DT, KLV>(m, p, queries, Q, k, use_min, min_score, score_mode, row_stride, row_offset,      \
                                out_scores, out_rows, uncertified, qflags, part_s, part_o, thr_s, thr_o, st)
    switch (p.cfg.KL) {
        case 8: GO(8);
        case 16: GO(16);
        case 32: GO(32);
        default: GO(64);
    }
#undef GO
}

}  // namespace

extern "C" size_t vm_topk_workspace_bytes(const vm_memory *m, int Q, int k) {
    if (!m || Q <= 0 || k <= 0 || k > 58) return 0;
    const ScanPlan p = make_plan(m, Q, k);
    return 2 * p.part_bytes + vm_align_up((size_t)p.q_pad * 8, 256) + (p.emit ? vm_topk_emit_workspace_bytes(p.q_pad) : 0) +
           256;
}

extern "C" int vm_topk_cosine(vm_memory *m, const void *queries, int Q, int k, int use_min_score,
                              double min_score, int score_mode, int64_t row_stride, int64_t row_offset,
                              double *out_scores, int64_t *out_rows, int32_t *out_uncertified,
                              int32_t *out_query_flags, void *workspace, size_t workspace_bytes, void *stream) {
    if (!m) return VM_ERR_INVALID;
    vm_ctx *ctx = m->ctx;
    if (!queries || !out_scores || !out_rows || Q <= 0 || k <= 0)
        return vm_fail(ctx, VM_ERR_INVALID, "vm_topk_cosine: bad arguments (Q=%d k=%d)", Q, k);
    if (k > 58)
        return vm_fail(ctx, VM_ERR_UNSUPPORTED, "vm_topk_cosine: k=%d > 58; use vm_topk_cosine_exact", k);
    if (score_mode != VM_SCORE_RAW && score_mode != VM_SCORE_UNIT_INTERVAL)
        return vm_fail(ctx, VM_ERR_INVALID, "bad score_mode %d", score_mode);
    const ScanPlan p = make_plan(m, Q, k);
    const size_t need = 2 * p.part_bytes + vm_align_up((size_t)p.q_pad * 8, 256) +
                        (p.emit ? vm_topk_emit_workspace_bytes(p.q_pad) : 0);
    if (!workspace || workspace_bytes < need)
        return vm_fail(ctx, VM_ERR_NOMEM, "vm_topk_cosine: workspace %zu < %zu", workspace_bytes, need);
    if (((uintptr_t)workspace & 15) || ((uintptr_t)queries & 15))
        return vm_fail(ctx, VM_ERR_INVALID, "vm_topk_cosine: pointers must be 16-byte aligned");
    float *part_s = (float *)workspace;
    int *part_o = (int *)((char *)workspace + p.part_bytes);
    float *thr_s = (float *)((char *)workspace + 2 * p.part_bytes);
    int *thr_o = (int *)(thr_s + p.q_pad);
    hipStream_t st = (hipStream_t)stream;
    if (m->dtype == VM_F16)
        return run_topk<VM_F16>(m, p, queries, Q, k, use_min_score, min_score, score_mode, row_stride,
                                row_offset, out_scores, out_rows, out_uncertified, out_query_flags, part_s, part_o, thr_s, thr_o,
                                st);
    return run_topk<VM_BF16>(m, p, queries, Q, k, use_min_score, min_score, score_mode, row_stride, row_offset,
                             out_scores, out_rows, out_uncertified, out_query_flags, part_s, part_o, thr_s, thr_o, st);
}

extern "C" int vm_topk_merge(vm_ctx *ctx, const double *scores, const int64_t *rows, int parts, int Q, int k,
                             double *out_scores, int64_t *out_rows, void *stream) {
    if (!ctx || !scores || !rows || !out_scores || !out_rows || parts <= 0 || Q <= 0 || k <= 0)
        return vm_fail(ctx, VM_ERR_INVALID, "vm_topk_merge: bad arguments");
    vm_prof_scope prof(ctx, VM_PROF_TOPK_MERGE, (hipStream_t)stream);
    topk_merge_kernel<<<Q, 256, 0, (hipStream_t)stream>>>(scores, rows, parts, Q, k, out_scores, out_rows);
    VM_LAUNCH_CHECK(ctx);
    return VM_OK;
}
