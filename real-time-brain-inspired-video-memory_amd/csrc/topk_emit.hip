// Cosine top-k scan for MANY queries per launch (Q >= 49): query-stationary, emit-only.
//
// Same contract and call sites as topk.hip (src/components/pre_llm_injector.py:346-388, the per-query loop over the
// whole memory; src/pipeline/retriever_hybrid.py:293-306).  The list scan of topk.hip keeps a sorted candidate list per
// lane and per query tile in registers; with 64+ queries per pass those lists fill the register file (two waves per
// SIMD) and the sorted inserts, not HBM, set the pace (Q = 256 over 1M x 768: 1.0 ms against a 0.19 ms HBM pass).
// Here instead:
//   * every wave keeps ITS 16 queries in registers as the MFMA B operand for the whole launch (D/32 fragments of 4
//     VGPRs: 96 registers at D = 768), 8 waves = 128 queries per workgroup ("superblock");
//   * the memory rows stream HBM -> LDS exactly once per superblock (LDS-DMA, 32-row tiles, chunk-XOR swizzle on the
//     source address and on the ds_read_b128 address, three tiles in flight) and are the MFMA A operand of all waves;
//   * a score is compared with the query's CUT - the KL-th best score of a sample of the rows, produced by the
//     existing list scan over the first rows (topk.hip SAMPLE pass) - and only scores at or above the cut are EMITTED
//     as candidates (expected rows * KL / sample ~ 1 k per query).  No per-lane lists, no sorted inserts.
//     Emission goes through a wave-private LDS buffer (ballot + prefix count) that is flushed 64 entries at a time
//     with one atomic per entry, so the hot loop contains no returning global atomic;
//   * topk_compact_kernel turns each query's candidate buffer into one sorted list of the KL best (bitwise radix
//     select + rank count of the survivors), which topk.hip's finalize kernel re-scores exactly and certifies as
//     before.  A query with more candidates than the buffer holds is marked and goes through the exhaustive redo
//     (topk_exact.hip): never a wrong answer, only a slower one.
#include "vm_internal.h"

#include <climits>
#include <type_traits>
#include <utility>

namespace {

constexpr int EM_THREADS = 512;
constexpr int EM_WAVES = EM_THREADS / 64;
constexpr int EM_ROWS = 32;        // rows per tile
constexpr int EM_QPB = 128;        // queries per superblock (16 per wave)
constexpr int EM_WBUF = 128;       // entries of a wave's LDS emission buffer (flushed when more than 64 are pending)
constexpr int CP_THREADS = 256;
constexpr int CP_PER_THREAD = VM_EMIT_CAP / CP_THREADS;

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

__device__ __forceinline__ bool better(float s1, int o1, float s2, int o2) {
    return s1 > s2 || (s1 == s2 && o1 < o2);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS accesses of the hot loop are inline asm: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of every LDS
// access it can see while an LDS-DMA is in flight (an LDS-DMA is a pending LDS write to its alias analysis), which
// drained the three-tile prefetch once per tile (3 us per tile instead of 1).  The hand-placed counted waits below
// order every read behind the DMA that feeds it: vmcnt before the tile's barrier, lgkmcnt before each MFMA batch.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // a native vector: asm operands must be register values
template <int OFF>
__device__ __forceinline__ void lds_read_b128(u32x4 &dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// wait for the batch (a, b, c, d): the statement names what it guards, so no consumer is scheduled above it
#define VM_WAIT_LGKM4(N, a, b, c, d) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
__device__ __forceinline__ void lds_write_b32(unsigned addr, unsigned v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// KS = D / 128.  STAGES tiles of [32 rows][D] 16-bit + 32 reciprocal norms each, then the emission buffers.
// NG = 16-query groups per wave: 1 (<= 128 queries per superblock: HBM-bound) or 2 (<= 256: every A fragment read from
// LDS feeds two MFMAs; with one group per wave the eight waves' A reads - 384 KB per tile - saturate the LDS).
// thr_s == null: no cut, every score is emitted (the first pass of the cut cascade, over a few hundred rows).
template <int DT, int KS, int NG, bool DEEP>
__global__ void __launch_bounds__(EM_THREADS, 1)
    topk_emit_kernel(const uint16_t *__restrict__ mem, const float *__restrict__ rnorm,
                     const uint16_t *__restrict__ queries, const int64_t *__restrict__ d_total, int64_t cap, int ring,
                     int Q, const float *__restrict__ thr_s, const int *__restrict__ thr_o, int *__restrict__ cand_cnt,
                     float *__restrict__ cand_s, int *__restrict__ cand_o, int nsuper, int64_t row_begin,
                     int64_t row_limit, int nt_rows) {
    using E = vm_elem<DT>;
    using vec8 = typename E::vec8;
    constexpr int D = 128 * KS;
    constexpr int ROW_BYTES = 2 * D;
    constexpr int TILE_BYTES = EM_ROWS * ROW_BYTES;          // 8 KiB * KS
    constexpr int STAGE_BYTES = TILE_BYTES + 256;            // + 32 fp32 reciprocal norms (256-byte slot)
    constexpr int STAGES = KS <= 6 ? 3 : 2;
    constexpr int PIECES = TILE_BYTES / 1024;                // 1 KiB LDS-DMA pieces per tile
    constexpr int PPW = (PIECES + EM_WAVES - 1) / EM_WAVES;  // pieces a wave issues per tile (+1 norm piece: wave 0)
    constexpr int KSTEPS = D / 32;
    // two batches of four A fragments in flight per wave where the registers allow it; NG = 2 (2 x 16 KS registers of
    // queries) and D = 1024 run one batch at a time and rely on the partner wave to cover the LDS round trip
    constexpr bool DBUF = NG == 1 && KS <= 6;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *stage0 = smem;
    char *ebuf = smem + STAGES * STAGE_BYTES;                // [waves][EM_WBUF] {f32 score, i32 order, i32 query}

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, h = lane >> 4;
    // workgroup -> (row block bx of nbx, superblock by): the superblocks of one row block are neighbours on ONE XCD
    // (workgroup id % 8 picks the XCD), so one of them pulls a tile from HBM / MALL and the others hit that L2
    int v = blockIdx.x;
    const int total_wg = gridDim.x;
    if ((total_wg & 7) == 0) v = (blockIdx.x & 7) * (total_wg >> 3) + (blockIdx.x >> 3);
    const int nbx = total_wg / nsuper;
    const int bx = v / nsuper, by = v - bx * nsuper;
    constexpr int QPB = EM_QPB * NG;
    const int q0 = by * QPB;
    const int nq = Q - q0 < QPB ? Q - q0 : QPB;              // queries of this superblock
    const int G = (nq + 15) >> 4;                            // 16-query groups in use (1 .. 8 NG)
    // NG = 1, G <= 4: two waves per group, one 16-row block of every tile each; otherwise a wave takes both row blocks
    // of a tile for its NG groups
    const bool paired = NG == 1 && G <= 4;
    const int g = paired ? (wave >> 1) : wave * NG;          // first group of this wave
    const int rb0 = paired ? (wave & 1) : 0, rb1 = paired ? (wave & 1) + 1 : 2;
    const bool active = g < G;

    // the wave's queries as B fragments: lane (r16, h) holds query r16 of a group, elements 32 s + 8 h .. + 7 of every
    // k-step; the cut of that query; padded lanes get +inf (nothing is "at or above" it: they never emit)
    vec8 bq[NG][KSTEPS];
    int myq[NG];
    float ts[NG];
    int to[NG];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        myq[u] = q0 + 16 * (g + u) + r16;
        const bool have_q = active && g + u < G && myq[u] < Q;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            uint4 w = make_uint4(0, 0, 0, 0);
            if (have_q) w = *reinterpret_cast<const uint4 *>(queries + (size_t)myq[u] * D + 32 * s + 8 * h);
            bq[u][s] = __builtin_bit_cast(vec8, w);
        }
        ts[u] = have_q ? (thr_s ? thr_s[myq[u]] : -INFINITY) : INFINITY;
        to[u] = have_q ? (thr_s ? thr_o[myq[u]] : INT_MAX) : -1;
    }
    // retire these ordinary loads HERE: left pending, hipcc waits for them with vmcnt(0) at their first use inside the
    // tile loop - in every iteration, which also drains the LDS-DMA prefetch
#pragma unroll
    for (int u = 0; u < NG; ++u) {
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) asm volatile("" ::"v"(bq[u][s]));
        asm volatile("" ::"v"(ts[u]), "v"(to[u]));
    }

    RingView rv = ring_view(*d_total, cap, ring);
    const DenseRange dense = dense_newest(rv);   // the rows of the dense pass (thr_s == null); skipped by the others
    if (!thr_s) {
        row_begin = dense.d0;
        row_limit = dense.d1;
    }
    if (rv.n > row_limit) rv.n = row_limit;  // cut cascade: a pass scans the physical slots [row_begin, row_limit) only
    const int64_t tile0 = row_begin / EM_ROWS;                                  // row_begin is a multiple of EM_ROWS
    const int64_t ntiles = rv.n > row_begin ? (rv.n + EM_ROWS - 1) / EM_ROWS - tile0 : 0;
    const int64_t my_tiles = bx < ntiles ? (ntiles - bx + nbx - 1) / nbx : 0;   // tiles tile0 + bx, + nbx, ...

    // LDS-DMA of one tile: piece p covers LDS bytes [1024 p, 1024 p + 1024) of the stage; lane -> (row, chunk') of the
    // linear image; the source chunk is chunk' ^ (row & 15) (swizzle on the source address, the reads apply it again)
    auto stage_piece = [&](int64_t tile, int buf, int i) {   // i-th of this wave's PPW row pieces of a tile
        const int p = wave * PPW + i;
        int ln = lane;
        asm volatile("" : "+v"(ln));  // recompute the per-lane source offset at every call: hoisted out of the tile loop
                                      // the PPW 64-bit offsets cost 2 PPW registers the two-group variant does not have
        const int off = p * 1024 + ln * 16;
        const int row = off / ROW_BYTES, cp = (off - row * ROW_BYTES) >> 4;
        const int c = cp ^ (row & 15);
        // one superblock = every row byte is read exactly once in this launch: non-temporal (aux 2); with several, the
        // superblocks of a row block share the tile through their XCD's L2 and keep the default policy
        if (nt_rows)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(mem + ((size_t)tile * EM_ROWS + row) * D + c * 8),
                                             (lds_ptr_t)(stage0 + buf * STAGE_BYTES + p * 1024), 16, 0, 2);
        else
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(mem + ((size_t)tile * EM_ROWS + row) * D + c * 8),
                                             (lds_ptr_t)(stage0 + buf * STAGE_BYTES + p * 1024), 16, 0, 0);
    };
    auto stage_norm = [&](int64_t tile, int buf) {  // last wave only: 64 reciprocal norms (this tile's 32 + 32 more)
        int64_t ri = tile * EM_ROWS + lane;
        const int64_t last = ((cap + 63) / 64) * 64 - 1;   // the allocation is padded to 64 rows
        if (ri > last) ri = last;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(rnorm + ri),
                                         (lds_ptr_t)(stage0 + buf * STAGE_BYTES + TILE_BYTES), 4, 0, 0);
    };
    auto stage_tile = [&](int64_t tile, int buf) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) stage_piece(tile, buf, i);
        if (wave == EM_WAVES - 1) stage_norm(tile, buf);
    };
    static_assert(PIECES % EM_WAVES == 0, "every wave issues exactly PPW row pieces per tile (the counted waits rely on it)");

    // emission buffer of this wave
    float *eb_s = reinterpret_cast<float *>(ebuf + wave * EM_WBUF * 12);
    int *eb_o = reinterpret_cast<int *>(eb_s + EM_WBUF);
    int *eb_q = eb_o + EM_WBUF;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;          // LDS byte address of the stages
    const unsigned eb0 = (unsigned)(uintptr_t)(lds_ptr_t)eb_s;           // ... and of this wave's emission buffer
    int pending = 0;  // wave-uniform
    auto flush = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the asm ds_writes above have landed
        for (int i = lane; i < pending; i += 64) {
            const int q = eb_q[i];
            const int slot = atomicAdd(&cand_cnt[q], 1);
            if (slot < VM_EMIT_CAP) {
                cand_s[(size_t)q * VM_EMIT_CAP + slot] = eb_s[i];
                cand_o[(size_t)q * VM_EMIT_CAP + slot] = eb_o[i];
            }
        }
        pending = 0;
    };

    // prologue: DEEP fills every stage, the shallow schedule leaves one free for the tile staged during the first compute
#pragma unroll
    for (int s = 0; s < (DEEP ? STAGES : STAGES - 1); ++s)
        if (s < my_tiles) stage_tile(tile0 + bx + (int64_t)s * nbx, s);

    for (int64_t it = 0; it < my_tiles; ++it) {
        const int64_t tile = tile0 + bx + it * nbx;
        const int buf = (int)(it % STAGES);
        bool do_stage, spread;
        int64_t ntile;
        int nbuf;
        if constexpr (DEEP) {
            // A workgroup of a bandwidth-bound scan computes ~1 k cycles per tile and waits ~4 k for the next one: what
            // matters is bytes in flight.  A second barrier right behind the compute frees the tile's stage at once, and
            // the tile STAGES ahead goes into it before the wait: all STAGES stages are in flight while the CU waits
            // (the shallow schedule below has STAGES - 1).
            const int64_t younger = my_tiles - 1 - it < STAGES - 1 ? my_tiles - 1 - it : STAGES - 1;
            if (younger >= 2) {
                if (wave == EM_WAVES - 1) wait_vmcnt<2 * PPW + 2>(); else wait_vmcnt<2 * PPW>();
            } else if (younger == 1) {
                if (wave == EM_WAVES - 1) wait_vmcnt<PPW + 1>(); else wait_vmcnt<PPW>();
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();  // every wave's pieces of this tile have landed
            do_stage = false;
            spread = false;
            ntile = tile + (int64_t)STAGES * nbx;
            nbuf = buf;
        } else {
            // retire this tile's DMA: younger ones are those of the (STAGES - 2) tiles staged after it
            if (STAGES == 3 && it + 1 < my_tiles) {
                // exactly the next tile's pieces may stay in flight: PPW per wave, + the norm piece on the last wave
                if (wave == EM_WAVES - 1) wait_vmcnt<PPW + 1>(); else wait_vmcnt<PPW>();
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();  // every wave's pieces of this tile have landed; everyone is done with tile it-1
            // LDS-DMA of tile it + STAGES - 1: an idle wave issues its pieces at once; a computing wave spreads them
            // over its MFMA batches below (an LDS-DMA instruction takes ~100 cycles to issue: issued as one block by
            // all eight waves right after the barrier, nobody fed the matrix pipe for ~1 k cycles per tile)
            do_stage = it + STAGES - 1 < my_tiles;
            ntile = tile + (int64_t)(STAGES - 1) * nbx;
            nbuf = (int)((it + STAGES - 1) % STAGES);
            // (spreading them also over the 2 KS batches of an un-paired wave cost 15 %: those waves are LDS-read-bound
            // and every extra instruction between their MFMA batches shows)
            spread = paired && active;
            if (do_stage && !spread) stage_tile(ntile, nbuf);
        }
        if (active) {
        const unsigned tb = lds0 + buf * STAGE_BYTES;  // LDS byte address of this tile's image
        for (int rb = rb0; rb < rb1; ++rb) {
            // A fragment of k-step s = 4 m + t: logical chunk c = 16 m + 4 t + h of row (16 rb + r16), stored at chunk
            // 16 m + ((4 t + h) ^ r16): four per-lane bases (t = 0..3), the m-th 256-byte group is an immediate offset
            const unsigned rowa = tb + (16 * rb + r16) * ROW_BYTES;
            unsigned base[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) base[t] = rowa + (((4 * t + h) ^ r16) << 4);
            u32x4 rn4;
            if constexpr (NG == 1) lds_read_b128<0>(rn4, tb + TILE_BYTES + (16 * rb + 4 * h) * 4);
            u32x4 e0, e1, e2, e3, o0, o1, o2, o3;  // even / odd batch of four A fragments
            f32x4 accs[NG];
#pragma unroll
            for (int u = 0; u < NG; ++u) accs[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#define VM_ISSUE4(M, a, b, c, d)                 \
    lds_read_b128<256 * (M)>(a, base[0]);        \
    lds_read_b128<256 * (M)>(b, base[1]);        \
    lds_read_b128<256 * (M)>(c, base[2]);        \
    lds_read_b128<256 * (M)>(d, base[3]);
#define VM_MMA4(M, a, b, c, d)                                                     \
    _Pragma("unroll") for (int u = 0; u < NG; ++u) {                             \
        accs[u] = E::mfma16(__builtin_bit_cast(vec8, a), bq[u][4 * (M) + 0], accs[u]); \
        accs[u] = E::mfma16(__builtin_bit_cast(vec8, b), bq[u][4 * (M) + 1], accs[u]); \
        accs[u] = E::mfma16(__builtin_bit_cast(vec8, c), bq[u][4 * (M) + 2], accs[u]); \
        accs[u] = E::mfma16(__builtin_bit_cast(vec8, d), bq[u][4 * (M) + 3], accs[u]); \
    }
            // batch m + 1 is issued before batch m is waited for (counted lgkmcnt(4)): one batch always in flight
            // NG = 2 has no registers for a second batch (2 x 96 hold the queries): the partner wave's 8 MFMAs per batch
            // cover this wave's LDS round trip instead
#define VM_STEP(M, ca, cb, cc, cd, na, nb, nc, nd)                       \
    if constexpr ((M) < KS) {                                             \
        if constexpr (DBUF && (M) + 1 < KS) {                             \
            VM_ISSUE4((M) + 1, na, nb, nc, nd)                            \
            VM_WAIT_LGKM4(4, ca, cb, cc, cd);                             \
        } else {                                                          \
            if constexpr (!DBUF && (M) > 0) { VM_ISSUE4(M, ca, cb, cc, cd) } \
            VM_WAIT_LGKM4(0, ca, cb, cc, cd);                             \
        }                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                \
        if (do_stage && spread) {                                         \
            stage_piece(ntile, nbuf, (M));                                \
            if (wave == EM_WAVES - 1 && (M) == KS - 1) stage_norm(ntile, nbuf); \
        }                                                                 \
        VM_MMA4(M, ca, cb, cc, cd)                                        \
    }
            VM_ISSUE4(0, e0, e1, e2, e3)
            if constexpr (DBUF) {
                VM_STEP(0, e0, e1, e2, e3, o0, o1, o2, o3)
                VM_STEP(1, o0, o1, o2, o3, e0, e1, e2, e3)
                VM_STEP(2, e0, e1, e2, e3, o0, o1, o2, o3)
                VM_STEP(3, o0, o1, o2, o3, e0, e1, e2, e3)
                VM_STEP(4, e0, e1, e2, e3, o0, o1, o2, o3)
                VM_STEP(5, o0, o1, o2, o3, e0, e1, e2, e3)
                VM_STEP(6, e0, e1, e2, e3, o0, o1, o2, o3)
                VM_STEP(7, o0, o1, o2, o3, e0, e1, e2, e3)
            } else {
                VM_STEP(0, e0, e1, e2, e3, e0, e1, e2, e3)
                VM_STEP(1, e0, e1, e2, e3, e0, e1, e2, e3)
                VM_STEP(2, e0, e1, e2, e3, e0, e1, e2, e3)
                VM_STEP(3, e0, e1, e2, e3, e0, e1, e2, e3)
                VM_STEP(4, e0, e1, e2, e3, e0, e1, e2, e3)
                VM_STEP(5, e0, e1, e2, e3, e0, e1, e2, e3)
                VM_STEP(6, e0, e1, e2, e3, e0, e1, e2, e3)
                VM_STEP(7, e0, e1, e2, e3, e0, e1, e2, e3)
            }
#undef VM_STEP
#undef VM_MMA4
#undef VM_ISSUE4
            if constexpr (NG == 2) {  // no register to keep the norms live across the batches: fetched here
                lds_read_b128<0>(rn4, tb + TILE_BYTES + (16 * rb + 4 * h) * 4);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rn4));
            }
            // acc[j] = <row tile*32 + 16 rb + 4 h + j , query myq>; rn4 landed before the first batch (in-order queue)
            // (elements copied out first: __builtin_bit_cast on an ext-vector ELEMENT expression reads element 0 for
            // every element with this hipcc - all four rows were scaled by rn4.x)
            const unsigned rn0 = rn4[0], rn1 = rn4[1], rn2 = rn4[2], rn3 = rn4[3];
            const float rnv[4] = {__builtin_bit_cast(float, rn0), __builtin_bit_cast(float, rn1),
                                  __builtin_bit_cast(float, rn2), __builtin_bit_cast(float, rn3)};
            const int64_t p0 = tile * EM_ROWS + 16 * rb + 4 * h;
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                // fast reject of the whole 4 x 64 score block with one compare: the best of the lane's four scaled
                // scores against the cut (rows past the end of the memory hold zeros or old rows: they can only make
                // the test pass needlessly; validity, order and the tie rule are applied in the rare slow path)
                const float s0 = accs[u][0] * rnv[0], s1 = accs[u][1] * rnv[1], s2 = accs[u][2] * rnv[2],
                            s3 = accs[u][3] * rnv[3];
                const float sv[4] = {s0, s1, s2, s3};
                if (!thr_s) {
                    // no cut (first pass of the cascade): EVERY score is a candidate - written straight to slot =
                    // physical index of its row, no counter, no atomics (through the emission buffer this pass took 75 us
                    // for 1024 rows: one flush with 64 atomics per ballot)
                    const bool have = ts[u] != INFINITY;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int64_t p = p0 + j;
                        int64_t o64 = p - rv.head;
                        if (o64 < 0) o64 += rv.cap;
                        if (have && p < rv.n && p - dense.d0 < VM_EMIT_CAP) {  // slot = index inside the dense range
                            cand_s[(size_t)myq[u] * VM_EMIT_CAP + (p - dense.d0)] = sv[j];
                            cand_o[(size_t)myq[u] * VM_EMIT_CAP + (p - dense.d0)] = (int)o64;
                        }
                    }
                    continue;
                }
                const float best4 = fmaxf(fmaxf(s0, s1), fmaxf(s2, s3));
                if (__ballot(best4 >= ts[u]) == 0ull) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t p = p0 + j;
                    int64_t o64 = p - rv.head;
                    if (o64 < 0) o64 += rv.cap;
                    const int o = (int)o64;
                    const float sc = sv[j];
                    const bool pass = p < rv.n && !(p >= dense.d0 && p < dense.d1) &&  // (dense rows are candidates already)
                                      !better(ts[u], to[u], sc, o);                    // at or above the cut
                    const unsigned long long m = __ballot(pass);
                    if (m) {  // wave-uniform
                        if (pending > EM_WBUF - 64) flush();
                        if (pass) {
                            const int idx = pending + __builtin_amdgcn_mbcnt_hi(
                                                          (unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                            lds_write_b32(eb0 + idx * 4, __builtin_bit_cast(unsigned, sc));
                            lds_write_b32(eb0 + (EM_WBUF + idx) * 4, (unsigned)o);
                            lds_write_b32(eb0 + (2 * EM_WBUF + idx) * 4, (unsigned)myq[u]);
                        }
                        pending += __popcll(m);
                    }
                }
            }
        }
        }  // active
        if constexpr (DEEP) {
            __builtin_amdgcn_s_barrier();  // everyone is done reading this stage
            if (it + STAGES < my_tiles) stage_tile(ntile, nbuf);
        }
    }
    if (pending) flush();
    if (!thr_s && bx == 0 && active && tid < 64 * EM_WAVES) {  // dense pass: candidate count = rows scanned
#pragma unroll
        for (int u = 0; u < NG; ++u)
            if (ts[u] != INFINITY && h == 0)
                cand_cnt[myq[u]] = (int)(dense.d1 - dense.d0);   // <= 4,095 by construction
    }
}

// One block per query: candidate buffer -> ONE sorted list of the KL best (score desc, order asc), in the list layout
// topk.hip's finalize kernel reads (nblk = 1), and / or the query's next CUT = the KL-th best (cut cascade).
// The KL-th largest score key is found by a 4-pass radix select (8 bits per pass: LDS histogram of the keys that match
// the prefix, Hillis-Steele suffix sums over the 256 bins), then everything at or above it - all ties included - is
// ranked by (score desc, order asc).  (Two earlier versions - a bitwise search with a block reduction per bit, KL
// rounds of block arg-best - spent 18-24 us per launch in chains of dependent cross-lane shuffles.)
// mark[q] is SET (never cleared: the caller zeroes it once per search) when the buffer overflowed or more ties sit at
// the KL-th place than are ranked here: finalize then flags the query for the exhaustive redo.
// seed != 0 (every pass of the cut cascade but the last): the ranked list is also written back to the head of the
// query's candidate buffer and cand_cnt[q] set to its length, so the next pass scans only the rows this one did not
// see and APPENDS to it - the KL best of a subset plus everything at or above their KL-th in the rest contain the KL
// best of the union.
constexpr int CP_SURV = 512;
__global__ void __launch_bounds__(CP_THREADS)
    topk_compact_kernel(int *__restrict__ cand_cnt, float *__restrict__ cand_s, int *__restrict__ cand_o, int KL,
                        float *__restrict__ part_s, int *__restrict__ part_o, int *__restrict__ mark,
                        float *__restrict__ cut_s, int *__restrict__ cut_o, int seed) {
    __shared__ int hist[2][CP_THREADS];
    __shared__ float sv_s[CP_SURV];
    __shared__ int sv_o[CP_SURV];
    __shared__ int nsurv, sel_digit, sel_rem;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int cnt = cand_cnt[q];
    const int C = cnt < VM_EMIT_CAP ? cnt : VM_EMIT_CAP;
    unsigned key[CP_PER_THREAD];
#pragma unroll
    for (int i = 0; i < CP_PER_THREAD; ++i) {
        const int c = tid + CP_THREADS * i;
        if (c < C) {
            const unsigned u = __builtin_bit_cast(unsigned, cand_s[(size_t)q * VM_EMIT_CAP + c]);
            const unsigned k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // order-preserving
            key[i] = k ? k : 1u;                                            // 0 is reserved for "no entry"
        } else {
            key[i] = 0u;
        }
    }
    if (tid == 0) nsurv = 0;
    const int want = C < KL ? C : KL;
    unsigned prefix = 0;       // the decided high bits of the want-th largest key
    int remaining = want;      // its rank among the keys that share the prefix
    if (want > 0) {
#pragma unroll 1
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            hist[0][tid] = 0;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < CP_PER_THREAD; ++i) {
                const bool in = key[i] != 0u && (pass == 0 || (key[i] >> (shift + 8)) == (prefix >> (shift + 8)));
                if (in) atomicAdd(&hist[0][(key[i] >> shift) & 255u], 1);
            }
            __syncthreads();
            // suffix sums S[t] = sum_{b >= t} hist[b], ping-pong between the two arrays
            int cur = 0;
#pragma unroll
            for (int d = 1; d < CP_THREADS; d <<= 1) {
                const int v = hist[cur][tid] + (tid + d < CP_THREADS ? hist[cur][tid + d] : 0);
                hist[cur ^ 1][tid] = v;
                __syncthreads();
                cur ^= 1;
            }
            const int mine = hist[cur][tid];
            const int above = tid + 1 < CP_THREADS ? hist[cur][tid + 1] : 0;
            if (mine >= remaining && above < remaining) {  // exactly one bin
                sel_digit = tid;
                sel_rem = remaining - above;
            }
            __syncthreads();
            prefix |= (unsigned)sel_digit << shift;
            remaining = sel_rem;
            __syncthreads();
        }
    }
    const unsigned tk = prefix;  // the want-th largest key
#pragma unroll
    for (int i = 0; i < CP_PER_THREAD; ++i) {
        if (want > 0 && key[i] != 0u && key[i] >= tk) {
            const int slot = atomicAdd(&nsurv, 1);
            if (slot < CP_SURV) {
                const int c = tid + CP_THREADS * i;
                sv_s[slot] = cand_s[(size_t)q * VM_EMIT_CAP + c];
                sv_o[slot] = cand_o[(size_t)q * VM_EMIT_CAP + c];
            }
        }
    }
    __syncthreads();
    const bool too_many = nsurv > CP_SURV;
    const int S = too_many ? CP_SURV : nsurv;
    float *ps = part_s + (size_t)q * KL;
    int *po = part_o + (size_t)q * KL;
    for (int i = tid; i < KL; i += CP_THREADS) {
        ps[i] = -INFINITY;
        po[i] = INT_MAX;
    }
    if (tid == 0 && cut_s && want < KL) {  // fewer than KL candidates: no valid cut
        cut_s[q] = -INFINITY;
        cut_o[q] = INT_MAX;
    }
    __syncthreads();
    for (int e = tid; e < S; e += CP_THREADS) {
        const float s = sv_s[e];
        const int o = sv_o[e];
        int rank = 0;
        for (int d = 0; d < S; ++d) rank += better(sv_s[d], sv_o[d], s, o) ? 1 : 0;
        if (rank < KL) {
            ps[rank] = s;
            po[rank] = o;
            if (seed) {  // every key of this block sits in registers and the survivors in LDS: the buffer is free
                cand_s[(size_t)q * VM_EMIT_CAP + rank] = s;
                cand_o[(size_t)q * VM_EMIT_CAP + rank] = o;
            }
            if (cut_s && rank == KL - 1) {
                cut_s[q] = s;
                cut_o[q] = o;
            }
        }
    }
    if (tid == 0) {
        if (mark && (cnt > VM_EMIT_CAP || too_many)) mark[q] = 1;
        if (seed) cand_cnt[q] = S < KL ? S : KL;  // ranks 0 .. min(S, KL) - 1 are all taken (distinct orders)
    }
}

template <int DT, int KS, int NG, bool DEEP>
int launch_emit(vm_memory *m, const void *queries, int Q, const float *thr_s, const int *thr_o, int *cand_cnt,
                float *cand_s, int *cand_o, int64_t row_begin, int64_t row_limit, hipStream_t st) {
    constexpr int D = 128 * KS;
    constexpr int STAGES = KS <= 6 ? 3 : 2;
    const size_t lds = (size_t)STAGES * (EM_ROWS * 2 * D + 256) + (size_t)EM_WAVES * EM_WBUF * 12;
    auto kern = topk_emit_kernel<DT, KS, NG, DEEP>;
    static unsigned long long attr = 0;   // one bit per device: the attribute belongs to the (kernel, device) pair
    if (!((attr >> (m->ctx->device & 63)) & 1ull)) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return vm_fail(m->ctx, VM_ERR_HIP, "emit LDS opt-in %zu: %s", lds, hipGetErrorString(e));
        attr |= 1ull << (m->ctx->device & 63);
    }
    const int nsuper = (Q + EM_QPB * NG - 1) / (EM_QPB * NG);
    int nbx = m->ctx->num_cus / nsuper;
    if (nbx < 1) nbx = 1;
    const int64_t rows = thr_s ? (m->cap < row_limit ? m->cap : row_limit) - row_begin : (int64_t)VM_EMIT_CAP;
    const int64_t ntiles = rows > 0 ? (rows + EM_ROWS - 1) / EM_ROWS : 1;
    if (nbx > ntiles) nbx = (int)ntiles;
    static const int nt_env = (int)VM_DEV_ENV("TOPK_NT", 1);
    // one superblock, and a pass that reads (nearly) the whole memory: every row byte is read once and never again
    const int nt_rows = nt_env && nsuper == 1 && row_limit >= m->cap && row_begin * 2 <= m->cap;
    vm_prof_scope prof(m->ctx, VM_PROF_TOPK_SCAN, st);
    kern<<<nbx * nsuper, EM_THREADS, lds, st>>>(m->rows, m->rnorm32, (const uint16_t *)queries, m->d_total, m->cap,
                                               m->ring, Q, thr_s, thr_o, cand_cnt, cand_s, cand_o, nsuper, row_begin,
                                               row_limit, nt_rows);
    VM_LAUNCH_CHECK(m->ctx);
    return VM_OK;
}

template <int DT, int KS>
int launch_emit_ng(vm_memory *m, const void *queries, int Q, const float *thr_s, const int *thr_o, int *cand_cnt,
                   float *cand_s, int *cand_o, int64_t row_begin, int64_t row_limit, hipStream_t st) {
    // bit 0: one group per wave (<= 128 queries), bit 1: two groups per wave.  Measured on 1 M x 768 (A/B on one box):
    // 256 queries 0.547 -> 0.528 ms with the deep schedule, 64 queries 0.312 -> 0.327 ms (a bandwidth-bound scan is not
    // short of bytes in flight; the second barrier only costs)
    static const int deep_env = (int)VM_DEV_ENV("EMIT_DEEP", 2);
    if constexpr (KS <= 6) {  // two query groups per wave need 2 x 16 KS registers for the queries alone (192 of 256 at
        if (Q > EM_QPB) {     // D = 768: checked spill-free with -Rpass-analysis=kernel-resource-usage)
            if (deep_env & 2)
                return launch_emit<DT, KS, 2, true>(m, queries, Q, thr_s, thr_o, cand_cnt, cand_s, cand_o, row_begin, row_limit, st);
            return launch_emit<DT, KS, 2, false>(m, queries, Q, thr_s, thr_o, cand_cnt, cand_s, cand_o, row_begin, row_limit, st);
        }
    }
    if (deep_env & 1)
        return launch_emit<DT, KS, 1, true>(m, queries, Q, thr_s, thr_o, cand_cnt, cand_s, cand_o, row_begin, row_limit, st);
    return launch_emit<DT, KS, 1, false>(m, queries, Q, thr_s, thr_o, cand_cnt, cand_s, cand_o, row_begin, row_limit, st);
}

}  // namespace

bool vm_topk_emit_supported(const vm_memory *m, int Q, int KL) {
    static const int env = (int)VM_DEV_ENV("TOPK_EMIT", 1);
    if (!env) return false;
    const int ks = m->D / 128;
    const bool d_ok = m->D % 128 == 0 && (ks == 1 || ks == 2 || ks == 4 || ks == 6 || ks == 8);
    // From 49 queries on, and - round 4 - for any query count once the per-lane lists would hold 32 or 64 entries
    // (k >= 11): a lane of the list scan sees ~120 scores of a 1 M-row memory, so a 32-entry sorted list is all warm-up
    // (75 of 120 scores pay a 256-instruction insert) and the scan is VALU-bound at 0.47 of the HBM rate (top-20 over
    // 1 M x 1024 bf16, BASELINE configs[2]); the emit scan compares a score with the query's cut and is HBM-bound.
    static const int kl32 = (int)VM_DEV_ENV("EMIT_KL32", 1);
    return d_ok && (Q >= 49 || (kl32 && KL >= 32)) && KL <= 64 && m->cap >= 65536;
}

size_t vm_topk_emit_workspace_bytes(int q_pad) {
    return vm_align_up((size_t)q_pad * 4, 256) * 2 + 2 * vm_align_up((size_t)q_pad * VM_EMIT_CAP * 4, 256);
}

// Scans the physical slots [row_begin, min(n, row_limit)) and APPENDS the candidates (cand_cnt: zero, or the seed
// count the previous pass's compact left).  thr_s == null: the DENSE pass - every score of the newest rows
// (dense_newest, vm_internal.h; computed on the device, row_begin / row_limit ignored), cand_cnt zero.
// Very many queries over many rows take the GEMM-class scan (topk_gscan.hip).
int vm_topk_emit_scan(vm_memory *m, const void *queries, int Q, int q_thr, const float *thr_s, const int *thr_o,
                      int *cand_cnt, float *cand_s, int *cand_o, int64_t row_begin, int64_t row_limit, hipStream_t st) {
    if (row_begin % EM_ROWS != 0)
        return vm_fail(m->ctx, VM_ERR_INVALID, "emit scan: row_begin %lld", (long long)row_begin);
    if (thr_s && row_begin % 256 == 0 &&
        vm_topk_gscan_supported(m, Q, (m->cap < row_limit ? m->cap : row_limit) - row_begin))
        return vm_topk_gscan(m, queries, Q, q_thr, thr_s, thr_o, cand_cnt, cand_s, cand_o, row_begin, row_limit, st);
    const int ks = m->D / 128;
#define GO(KSV)                                                                                                   \
    return m->dtype == VM_F16                                                                                     \
               ? launch_emit_ng<VM_F16, KSV>(m, queries, Q, thr_s, thr_o, cand_cnt, cand_s, cand_o, row_begin, row_limit, st) \
               : launch_emit_ng<VM_BF16, KSV>(m, queries, Q, thr_s, thr_o, cand_cnt, cand_s, cand_o, row_begin, row_limit, st)
    switch (ks) {
        case 1: GO(1);
        case 2: GO(2);
        case 4: GO(4);
        case 6: GO(6);
        case 8: GO(8);
        default: return vm_fail(m->ctx, VM_ERR_UNSUPPORTED, "emit scan: D=%d", m->D);
    }
#undef GO
}

int vm_topk_emit_compact(vm_memory *m, int Q, int KL, int *cand_cnt, float *cand_s, int *cand_o, float *part_s,
                         int *part_o, int *mark, float *cut_s, int *cut_o, int seed, hipStream_t st) {
    vm_prof_scope prof(m->ctx, VM_PROF_TOPK_FINALIZE, st);
    topk_compact_kernel<<<Q, CP_THREADS, 0, st>>>(cand_cnt, cand_s, cand_o, KL, part_s, part_o, mark, cut_s, cut_o, seed);
    VM_LAUNCH_CHECK(m->ctx);
    return VM_OK;
}
