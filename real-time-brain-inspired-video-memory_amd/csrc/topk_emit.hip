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
//     before.  A query with more candidates than the buffer holds, or more ties at the KL-th place than the kernel
//     ranks, is marked and goes through the exhaustive redo (topk_exact.hip): never a wrong answer, only a slower one.
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
constexpr int CP_SURV = 256;       // survivors the compact kernel ranks

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
template <int DT, int KS>
__global__ void __launch_bounds__(EM_THREADS, 1)
    topk_emit_kernel(const uint16_t *__restrict__ mem, const float *__restrict__ rnorm,
                     const uint16_t *__restrict__ queries, const int64_t *__restrict__ d_total, int64_t cap, int ring,
                     int Q, const float *__restrict__ thr_s, const int *__restrict__ thr_o, int *__restrict__ cand_cnt,
                     float *__restrict__ cand_s, int *__restrict__ cand_o, int nsuper) {
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
    const int q0 = by * EM_QPB;
    const int nq = Q - q0 < EM_QPB ? Q - q0 : EM_QPB;       // queries of this superblock
    const int G = (nq + 15) >> 4;                            // 16-query groups in use (1..8)
    // G <= 4: two waves per group, one 16-row block of every tile each; G > 4: one wave per group, both row blocks
    const bool paired = G <= 4;
    const int g = paired ? (wave >> 1) : wave;
    const int rb0 = paired ? (wave & 1) : 0, rb1 = paired ? (wave & 1) + 1 : 2;
    const bool active = g < G;
    const int myq = q0 + 16 * g + r16;                       // this lane's query
    const bool have_q = active && myq < Q;

    // the wave's 16 queries as B fragments: lane (r16, h) holds query r16, elements 32 s + 8 h .. + 7 of every k-step
    vec8 bq[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        uint4 u = make_uint4(0, 0, 0, 0);
        if (have_q) u = *reinterpret_cast<const uint4 *>(queries + (size_t)myq * D + 32 * s + 8 * h);
        bq[s] = __builtin_bit_cast(vec8, u);
    }
    const float ts = have_q ? thr_s[myq] : INFINITY;         // nothing is "at or above" +inf: padded lanes never emit
    const int to = have_q ? thr_o[myq] : -1;
    // retire these ordinary loads HERE: left pending, hipcc waits for them with vmcnt(0) at their first use inside the
    // tile loop - in every iteration, which also drains the LDS-DMA prefetch
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) asm volatile("" ::"v"(bq[s]));
    asm volatile("" ::"v"(ts), "v"(to));

    const RingView rv = ring_view(*d_total, cap, ring);
    const int64_t ntiles = (rv.n + EM_ROWS - 1) / EM_ROWS;
    const int64_t my_tiles = bx < ntiles ? (ntiles - bx + nbx - 1) / nbx : 0;   // tiles bx, bx + nbx, ...

    // LDS-DMA of one tile: piece p covers LDS bytes [1024 p, 1024 p + 1024) of the stage; lane -> (row, chunk') of the
    // linear image; the source chunk is chunk' ^ (row & 15) (swizzle on the source address, the reads apply it again)
    auto stage_tile = [&](int64_t tile, int buf) {
        char *dst = stage0 + buf * STAGE_BYTES;
        const uint16_t *tbase = mem + (size_t)tile * EM_ROWS * D;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave * PPW + i;
            if (p < PIECES) {
                const int off = p * 1024 + lane * 16;
                const int row = off / ROW_BYTES, cp = (off - row * ROW_BYTES) >> 4;
                const int c = cp ^ (row & 15);
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(tbase + (size_t)row * D + c * 8),
                                                 (lds_ptr_t)(dst + p * 1024), 16, 0, 0);
            }
        }
        if (wave == EM_WAVES - 1) {  // 64 reciprocal norms (this tile's 32 + 32 more; the allocation is padded to 64)
            int64_t ri = tile * EM_ROWS + lane;
            const int64_t last = ((cap + 63) / 64) * 64 - 1;
            if (ri > last) ri = last;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(rnorm + ri), (lds_ptr_t)(dst + TILE_BYTES), 4, 0, 0);
        }
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

    // prologue: STAGES - 1 tiles in flight
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < my_tiles) stage_tile(bx + (int64_t)s * nbx, s);

    for (int64_t it = 0; it < my_tiles; ++it) {
        const int64_t tile = bx + it * nbx;
        const int buf = (int)(it % STAGES);
        // retire this tile's DMA: younger ones are those of the (STAGES - 2) tiles staged after it
        if (STAGES == 3 && it + 1 < my_tiles) {
            // exactly the next tile's pieces may stay in flight: PPW per wave, + the norm piece on the last wave
            if (wave == EM_WAVES - 1) wait_vmcnt<PPW + 1>(); else wait_vmcnt<PPW>();
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();  // every wave's pieces of this tile have landed; everyone is done with tile it-1
        if (it + STAGES - 1 < my_tiles) stage_tile(tile + (int64_t)(STAGES - 1) * nbx, (int)((it + STAGES - 1) % STAGES));
        if (!active) continue;
        const unsigned tb = lds0 + buf * STAGE_BYTES;  // LDS byte address of this tile's image
        for (int rb = rb0; rb < rb1; ++rb) {
            // A fragment of k-step s = 4 m + t: logical chunk c = 16 m + 4 t + h of row (16 rb + r16), stored at chunk
            // 16 m + ((4 t + h) ^ r16): four per-lane bases (t = 0..3), the m-th 256-byte group is an immediate offset
            const unsigned rowa = tb + (16 * rb + r16) * ROW_BYTES;
            unsigned base[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) base[t] = rowa + (((4 * t + h) ^ r16) << 4);
            u32x4 rn4;
            lds_read_b128<0>(rn4, tb + TILE_BYTES + (16 * rb + 4 * h) * 4);
            u32x4 e0, e1, e2, e3, o0, o1, o2, o3;  // even / odd batch of four A fragments
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#define VM_ISSUE4(M, a, b, c, d)                 \
    lds_read_b128<256 * (M)>(a, base[0]);        \
    lds_read_b128<256 * (M)>(b, base[1]);        \
    lds_read_b128<256 * (M)>(c, base[2]);        \
    lds_read_b128<256 * (M)>(d, base[3]);
#define VM_MMA4(M, a, b, c, d)                                                  \
    acc = E::mfma16(__builtin_bit_cast(vec8, a), bq[4 * (M) + 0], acc);        \
    acc = E::mfma16(__builtin_bit_cast(vec8, b), bq[4 * (M) + 1], acc);        \
    acc = E::mfma16(__builtin_bit_cast(vec8, c), bq[4 * (M) + 2], acc);        \
    acc = E::mfma16(__builtin_bit_cast(vec8, d), bq[4 * (M) + 3], acc);
            // batch m + 1 is issued before batch m is waited for (counted lgkmcnt(4)): one batch always in flight
#define VM_STEP(M, ca, cb, cc, cd, na, nb, nc, nd)                       \
    if constexpr ((M) < KS) {                                             \
        if constexpr ((M) + 1 < KS) {                                     \
            VM_ISSUE4((M) + 1, na, nb, nc, nd)                            \
            VM_WAIT_LGKM4(4, ca, cb, cc, cd);                             \
        } else {                                                          \
            VM_WAIT_LGKM4(0, ca, cb, cc, cd);                             \
        }                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                \
        VM_MMA4(M, ca, cb, cc, cd)                                        \
    }
            VM_ISSUE4(0, e0, e1, e2, e3)
            VM_STEP(0, e0, e1, e2, e3, o0, o1, o2, o3)
            VM_STEP(1, o0, o1, o2, o3, e0, e1, e2, e3)
            VM_STEP(2, e0, e1, e2, e3, o0, o1, o2, o3)
            VM_STEP(3, o0, o1, o2, o3, e0, e1, e2, e3)
            VM_STEP(4, e0, e1, e2, e3, o0, o1, o2, o3)
            VM_STEP(5, o0, o1, o2, o3, e0, e1, e2, e3)
            VM_STEP(6, e0, e1, e2, e3, o0, o1, o2, o3)
            VM_STEP(7, o0, o1, o2, o3, e0, e1, e2, e3)
#undef VM_STEP
#undef VM_MMA4
#undef VM_ISSUE4
            // acc[j] = <row tile*32 + 16 rb + 4 h + j , query myq>; rn4 landed before the first batch (in-order queue)
            // (elements copied out first: __builtin_bit_cast on an ext-vector ELEMENT expression reads element 0 for
            // every element with this hipcc - all four rows were scaled by rn4.x)
            const unsigned rn0 = rn4[0], rn1 = rn4[1], rn2 = rn4[2], rn3 = rn4[3];
            const float rnv[4] = {__builtin_bit_cast(float, rn0), __builtin_bit_cast(float, rn1),
                                  __builtin_bit_cast(float, rn2), __builtin_bit_cast(float, rn3)};
            const int64_t p0 = tile * EM_ROWS + 16 * rb + 4 * h;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t p = p0 + j;
                int64_t o64 = p - rv.head;
                if (o64 < 0) o64 += rv.cap;
                const int o = (int)o64;
                const float sc = acc[j] * rnv[j];
                const bool pass = p < rv.n && !better(ts, to, sc, o);  // at or above the cut
                const unsigned long long m = __ballot(pass);
                if (m) {  // wave-uniform, rare
                    if (pending > EM_WBUF - 64) flush();
                    if (pass) {
                        const int idx = pending + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                                            __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                        lds_write_b32(eb0 + idx * 4, __builtin_bit_cast(unsigned, sc));
                        lds_write_b32(eb0 + (EM_WBUF + idx) * 4, (unsigned)o);
                        lds_write_b32(eb0 + (2 * EM_WBUF + idx) * 4, (unsigned)myq);
                    }
                    pending += __popcll(m);
                }
            }
        }
    }
    if (pending) flush();
}

// One block per query: candidate buffer -> ONE sorted list of the KL best (score desc, order asc), in the list layout
// topk.hip's finalize kernel reads (nblk = 1).  mark[q] = 1 when the buffer overflowed or the ties at the KL-th place
// outnumber what is ranked here: finalize then flags the query for the exhaustive redo.
__global__ void __launch_bounds__(CP_THREADS)
    topk_compact_kernel(const int *__restrict__ cand_cnt, const float *__restrict__ cand_s,
                        const int *__restrict__ cand_o, int KL, int q_pad, float *__restrict__ part_s,
                        int *__restrict__ part_o, int *__restrict__ mark) {
    __shared__ int wsum[CP_THREADS / 64];
    __shared__ float sv_s[CP_SURV];
    __shared__ int sv_o[CP_SURV];
    __shared__ int nsurv;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int cnt = cand_cnt[q];
    const int C = cnt < VM_EMIT_CAP ? cnt : VM_EMIT_CAP;
    unsigned key[CP_PER_THREAD];
    int ord[CP_PER_THREAD];
    float scv[CP_PER_THREAD];
#pragma unroll
    for (int i = 0; i < CP_PER_THREAD; ++i) {
        const int c = tid + CP_THREADS * i;
        const bool have = c < C;
        const float s = have ? cand_s[(size_t)q * VM_EMIT_CAP + c] : 0.f;
        const unsigned u = __builtin_bit_cast(unsigned, s);
        key[i] = have ? ((u & 0x80000000u) ? ~u : (u | 0x80000000u)) : 0u;  // order-preserving; 0 = no entry
        if (have && key[i] == 0u) key[i] = 1u;                              // (only -NaN maps to 0)
        ord[i] = have ? cand_o[(size_t)q * VM_EMIT_CAP + c] : INT_MAX;
        scv[i] = s;
    }
    if (tid == 0) nsurv = 0;
    auto block_count = [&](auto pred) {
        int c = 0;
#pragma unroll
        for (int i = 0; i < CP_PER_THREAD; ++i) c += pred(i) ? 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
        __syncthreads();
        if ((tid & 63) == 0) wsum[tid >> 6] = c;
        __syncthreads();
        return (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    };
    const int want = C < KL ? C : KL;
    unsigned tk = 0;  // largest key with count(key >= tk) >= want
    if (want > 0) {
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned cand = tk | (1u << bit);
            if (block_count([&](int i) { return key[i] >= cand; }) >= want) tk = cand;
        }
    }
    // survivors: everything above the KL-th key, plus ALL ties at it (their orders decide; ranked below)
    const int n_ge = want > 0 ? block_count([&](int i) { return key[i] != 0u && key[i] >= tk; }) : 0;
    const bool too_many = n_ge > CP_SURV;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < CP_PER_THREAD; ++i) {
        if (want > 0 && key[i] != 0u && key[i] >= tk) {
            const int slot = atomicAdd(&nsurv, 1);
            if (slot < CP_SURV) {
                sv_s[slot] = scv[i];
                sv_o[slot] = ord[i];
            }
        }
    }
    __syncthreads();
    const int S = nsurv < CP_SURV ? nsurv : CP_SURV;
    float *ps = part_s + (size_t)q * KL;
    int *po = part_o + (size_t)q * KL;
    (void)q_pad;
    for (int i = tid; i < KL; i += CP_THREADS) {
        ps[i] = -INFINITY;
        po[i] = INT_MAX;
    }
    __syncthreads();
    if (tid < S) {
        const float s = sv_s[tid];
        const int o = sv_o[tid];
        int rank = 0;
        for (int d = 0; d < S; ++d) rank += better(sv_s[d], sv_o[d], s, o) ? 1 : 0;
        if (rank < KL) {
            ps[rank] = s;
            po[rank] = o;
        }
    }
    if (tid == 0) mark[q] = (cnt > VM_EMIT_CAP || too_many) ? 1 : 0;
}

template <int DT, int KS>
int launch_emit(vm_memory *m, const void *queries, int Q, const float *thr_s, const int *thr_o, int *cand_cnt,
                float *cand_s, int *cand_o, hipStream_t st) {
    constexpr int D = 128 * KS;
    constexpr int STAGES = KS <= 6 ? 3 : 2;
    const size_t lds = (size_t)STAGES * (EM_ROWS * 2 * D + 256) + (size_t)EM_WAVES * EM_WBUF * 12;
    auto kern = topk_emit_kernel<DT, KS>;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return vm_fail(m->ctx, VM_ERR_HIP, "emit LDS opt-in %zu: %s", lds, hipGetErrorString(e));
        attr = true;
    }
    const int nsuper = (Q + EM_QPB - 1) / EM_QPB;
    int nbx = m->ctx->num_cus / nsuper;
    if (nbx < 1) nbx = 1;
    const int64_t ntiles = (m->cap + EM_ROWS - 1) / EM_ROWS;
    if (nbx > ntiles) nbx = (int)ntiles;
    vm_prof_scope prof(m->ctx, VM_PROF_TOPK_SCAN, st);
    kern<<<nbx * nsuper, EM_THREADS, lds, st>>>(m->rows, m->rnorm32, (const uint16_t *)queries, m->d_total, m->cap,
                                               m->ring, Q, thr_s, thr_o, cand_cnt, cand_s, cand_o, nsuper);
    VM_LAUNCH_CHECK(m->ctx);
    return VM_OK;
}

}  // namespace

bool vm_topk_emit_supported(const vm_memory *m, int Q, int KL) {
    static int env = -1;
    if (env < 0) {
        const char *e = getenv("VIDMEM_TOPK_EMIT");
        env = e ? atoi(e) : 1;
    }
    if (!env) return false;
    const int ks = m->D / 128;
    const bool d_ok = m->D % 128 == 0 && (ks == 1 || ks == 2 || ks == 4 || ks == 6 || ks == 8);
    return d_ok && Q >= 49 && KL <= 64 && m->cap >= 65536;
}

size_t vm_topk_emit_workspace_bytes(int q_pad) {
    return vm_align_up((size_t)q_pad * 4, 256) * 2 + 2 * vm_align_up((size_t)q_pad * VM_EMIT_CAP * 4, 256);
}

// cand_cnt must be zero when the scan starts (the caller memsets it on the stream)
int vm_topk_emit_scan(vm_memory *m, const void *queries, int Q, const float *thr_s, const int *thr_o, int *cand_cnt,
                      float *cand_s, int *cand_o, hipStream_t st) {
    const int ks = m->D / 128;
#define GO(KSV)                                                                                                  \
    return m->dtype == VM_F16 ? launch_emit<VM_F16, KSV>(m, queries, Q, thr_s, thr_o, cand_cnt, cand_s, cand_o, st) \
                              : launch_emit<VM_BF16, KSV>(m, queries, Q, thr_s, thr_o, cand_cnt, cand_s, cand_o, st)
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

int vm_topk_emit_compact(vm_memory *m, int Q, int KL, int q_pad, const int *cand_cnt, const float *cand_s,
                         const int *cand_o, float *part_s, int *part_o, int *mark, hipStream_t st) {
    vm_prof_scope prof(m->ctx, VM_PROF_TOPK_FINALIZE, st);
    topk_compact_kernel<<<Q, CP_THREADS, 0, st>>>(cand_cnt, cand_s, cand_o, KL, q_pad, part_s, part_o, mark);
    VM_LAUNCH_CHECK(m->ctx);
    return VM_OK;
}
