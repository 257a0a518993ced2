// Cosine top-k scan for MANY queries per launch (Q >= 129 over a large memory): the scan as a GEMM with an emit epilogue.
//
// Same contract and call sites as topk.hip / topk_emit.hip (src/components/pre_llm_injector.py:346-388: every query
// against every stored row; src/pipeline/retriever_hybrid.py:293-306).  This is the shape an 8-GPU step brings to every
// shard (BASELINE configs[3]: every rank scores ALL ranks' 8 x 880 = 7,040 queries against its 1,048,576 rows): 11.3
// TFLOP per search, MFMA-bound, 289 FLOP per row byte even if the rows came from HBM once per 256 queries.
//
// The query-stationary emit scan (topk_emit.hip) keeps 2 x 16 queries per wave in registers and reads every A fragment
// from LDS for two MFMAs: the LDS, not the matrix pipe, sets its pace (0.30 of the MFMA peak at 256 queries per pass).
// Here the scores of a 256-row x 256-query tile are accumulated the way the encoder's GEMMs are (gemm.hip,
// gemm256p_kernel: 8 waves as 2 x 4, 128 x 64 scores per wave = 8 x 4 accumulator blocks, 12 fragment reads per 32 MFMAs,
// both operands staged by buffer_load ... lds into a swizzled double buffer, four barrier phases per 64-deep K-tile,
// the K-tile stream running on across tile boundaries), and the epilogue, instead of storing anything, scales the
// scores by the rows' reciprocal norms, tests each 16 x 16 block against the queries' CUTS with one compare per lane and
// emits the few scores at or above the cut as candidates (wave-private LDS buffer, flushed 64 at a time) - the same
// candidate buffers, compact kernel, exact fp64 finalize and certification as the emit scan.
//
// Work split.  A tile is (row panel, query tile).  The 8 XCDs form a QX x RX grid (QX * RX = 8): an XCD owns
// ceil(query tiles / QX) query tiles - at most ~8, so their 393 KB each stay in its 4 MiB L2 for the whole launch - and
// every RX-th row panel; its 32 workgroups walk its tiles panel-major, so at any time the workgroups that share a row
// panel run side by side and the panel comes from HBM / MALL once per XCD.  (Which XCD a workgroup lands on is read
// from blockIdx % 8: observed placement, used for speed only.)
#include "vm_internal.h"

#include <climits>

namespace {

constexpr int GS_BK = 64;
constexpr int GS_HALF = 128 * GS_BK * 2;       // 128 rows x 64 k x 2 B = 16 KiB
constexpr int GS_STAGE = 8 * GS_HALF;          // [2 buffers][A half0 | A half1 | B half0 | B half1] = 128 KiB
constexpr int GS_TAB = 2048;                   // per tile parity: 256 reciprocal norms + 256 cut scores (fp32)
constexpr int GS_WBUF = 128;                   // entries of a wave's emission buffer
constexpr int GS_LDS = GS_STAGE + 2 * GS_TAB + 8 * GS_WBUF * 12;

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct GscanArgs {
    const uint16_t *mem;       // [cap_pad, D]
    const float *rnorm;        // [cap_pad]
    const uint16_t *queries;   // [Q, D]
    const int64_t *d_total;
    int64_t cap, cap_pad;
    int ring, D, Q, q_thr;     // q_thr: entries of thr_s / thr_o
    const float *thr_s;
    const int *thr_o;
    int *cand_cnt;
    float *cand_s;
    int *cand_o;
    int64_t row_begin, row_limit;   // physical slots [row_begin, min(n, row_limit)); row_begin % 256 == 0
    int QX;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_rd128(u32x4 &d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_rd32(unsigned &d, unsigned addr) {
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_wr32(unsigned addr, unsigned v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

template <int DT>
__global__ void __launch_bounds__(512, 1) topk_gscan_kernel(GscanArgs g) {
    using E = vm_elem<DT>;
    using vec8 = typename E::vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, h = lane >> 4;
    const int wr = wave >> 2, wc = wave & 3;   // wave row: 128 memory rows; wave column: 64 queries
    const int D = g.D, nk = D / GS_BK;

    // ---- which tiles are mine ----
    const int xcd = blockIdx.x & 7, cu = blockIdx.x >> 3, ncu = gridDim.x >> 3;
    const int QX = g.QX, RX = 8 / QX;
    const int qx = xcd % QX, rx = xcd / QX;
    const int nqt = (g.Q + 255) >> 8;
    const int qpx = (nqt + QX - 1) / QX;
    const int qt0 = qx * qpx;
    int nq_x = nqt - qt0;
    if (nq_x > qpx) nq_x = qpx;
    const RingView rv = ring_view(*g.d_total, g.cap, g.ring);
    const DenseRange dense = dense_newest(rv);   // already candidates (the cascade's dense pass): never emitted here
    const int64_t r_hi = rv.n < g.row_limit ? rv.n : g.row_limit;
    const int nrp = r_hi > g.row_begin ? (int)((r_hi - g.row_begin + 255) >> 8) : 0;
    const int np_x = nrp > rx ? (nrp - rx + RX - 1) / RX : 0;
    const int nt_x = nq_x > 0 ? np_x * nq_x : 0;
    int tile = cu;
    if (tile >= nt_x) return;  // the whole workgroup leaves together

    int64_t cur_row0, nxt_row0;   // first physical slot of the current / next tile's row panel
    int cur_q0, nxt_q0;           // first query of the current / next tile
    auto coords = [&](int tt, int64_t &row0, int &q0) {
        const int m = tt / nq_x, qi = tt - m * nq_x;
        row0 = g.row_begin + ((int64_t)(rx + RX * m) << 8);
        q0 = (qt0 + qi) << 8;
    };

    // ---- staging (gemm.hip, gemm256p_kernel): buffer_load_dwordx4 ... offen lds, 8 rows x 128 B per instruction.
    // One descriptor per operand and tile: base = the tile's first row, num_records = the bytes of the rows that exist,
    // so rows past the end of the memory / past the last query are out of range by their per-lane offset alone (the
    // wave-uniform soffset only ever carries the K-tile, 128 B x kt < one row).
    const int srow = lane >> 3, scp = lane & 7;
    const int chunk = scp ^ srow;
    unsigned voff_a[2], voff_b[2];
    int lds_a0[2], lds_b0[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int q = 2 * wave + u;
        const int arow0 = (q < 8 ? q * 8 : 128 + (q - 8) * 8);   // region Aa0 = rows {0..63, 128..191}; Aa1 = + 64
        const int brow0 = (q >> 2) * 64 + (q & 3) * 8;           // region Bb0 = rows {64 c + 0..31};    Bb1 = + 32
        voff_a[u] = (unsigned)(((arow0 + srow) * D + chunk * 8) * 2);
        voff_b[u] = (unsigned)(((brow0 + srow) * D + chunk * 8) * 2);
        lds_a0[u] = arow0 * 128;
        lds_b0[u] = 2 * GS_HALF + brow0 * 128;
    }
    const unsigned a64 = (unsigned)(64 * D * 2), b32 = (unsigned)(32 * D * 2);
    __amdgpu_buffer_rsrc_t rs_a, rs_b;
    auto set_sources = [&](int64_t row0, int q0) {
        int64_t vr = g.cap_pad - row0;
        if (vr > 256) vr = 256;
        int vq = g.Q - q0;
        if (vq > 256) vq = 256;
        rs_a = __builtin_amdgcn_make_buffer_rsrc((void *)(g.mem + (size_t)row0 * D), 0, (int)(vr * D * 2), 0x00020000);
        rs_b = __builtin_amdgcn_make_buffer_rsrc((void *)(g.queries + (size_t)q0 * D), 0, vq * D * 2, 0x00020000);
    };
    auto dma2 = [&](const __amdgpu_buffer_rsrc_t rs, const unsigned (&voff)[2], unsigned vextra, unsigned soff,
                    const int (&dst)[2], int extra, int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + buf * 4 * GS_HALF + dst[u] + extra), 16,
                                                     voff[u] + vextra, soff, 0, 0);
    };
    auto stage_Aa0 = [&](int kt, int buf) { dma2(rs_a, voff_a, 0u, kt * (GS_BK * 2), lds_a0, 0, buf); };
    auto stage_Aa1 = [&](int kt, int buf) { dma2(rs_a, voff_a, a64, kt * (GS_BK * 2), lds_a0, 64 * 128, buf); };
    auto stage_Bb0 = [&](int kt, int buf) { dma2(rs_b, voff_b, 0u, kt * (GS_BK * 2), lds_b0, 0, buf); };
    auto stage_Bb1 = [&](int kt, int buf) { dma2(rs_b, voff_b, b32, kt * (GS_BK * 2), lds_b0, 32 * 128, buf); };
    // the tile's table: 256 reciprocal norms (even waves) and 256 cut scores (odd waves), 1 KiB each, one LDS-DMA per
    // wave (the four waves of a parity write the same bytes: every wave issues the same number of operations, which
    // the counted waits below rely on)
    auto stage_tab = [&](int64_t row0, int q0, int par) {
        const void *src;
        int64_t nrec;
        if (wave & 1) {
            src = g.thr_s + q0;
            nrec = (int64_t)(g.q_thr - q0) * 4;
        } else {
            src = g.rnorm + row0;
            nrec = (g.cap_pad - row0) * 4;
        }
        if (nrec > 1024) nrec = 1024;
        const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, (int)nrec, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_t, (lds_ptr_t)(smem + GS_STAGE + par * GS_TAB + (wave & 1) * 1024), 16,
                                                 (unsigned)(lane * 16), 0, 0, 0);
    };

    const int sw0 = ((h ^ (r16 & 7)) << 4), sw1 = (((h + 4) ^ (r16 & 7)) << 4);
    const int a_base = (wr * 128 + r16) * 128;
    const int b_base = 2 * GS_HALF + (wc * 64 + r16) * 128;

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
    // acc[i][j][e] = <memory row (panel + 128 wr + 16 i + 4 h + e), query (q0 + 64 wc + 16 j + r16)>: rows are the
    // MFMA A operand, so the four scores of a lane share one query (one cut) - as in the emit scan
    // first: the first K-step of a tile starts from C = 0 inside the MFMA (inline constant) instead of 128 cleared
    // registers per wave and tile (zero_acc: ~2 % of a tile's time in vector moves)
    auto mma = [&](int ahalf, int bhalf, bool first) {
        __builtin_amdgcn_s_setprio(1);
        if (first) {
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
#define GS_BAR() __builtin_amdgcn_s_barrier()
#define GS_LGKM0()                                         \
    do {                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                 \
    } while (0)

    // ---- emission (topk_emit.hip): wave-private LDS buffer, ballot + prefix count, flushed 64 entries at a time ----
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    char *ebuf = smem + GS_STAGE + 2 * GS_TAB + wave * GS_WBUF * 12;
    float *eb_s = reinterpret_cast<float *>(ebuf);
    int *eb_o = reinterpret_cast<int *>(eb_s + GS_WBUF);
    int *eb_q = eb_o + GS_WBUF;
    const unsigned eb0 = (unsigned)(uintptr_t)(lds_ptr_t)ebuf;
    int pending = 0;  // wave-uniform
    auto flush = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = lane; i < pending; i += 64) {
            const int q = eb_q[i];
            const int slot = atomicAdd(&g.cand_cnt[q], 1);
            if (slot < VM_EMIT_CAP) {
                g.cand_s[(size_t)q * VM_EMIT_CAP + slot] = eb_s[i];
                g.cand_o[(size_t)q * VM_EMIT_CAP + slot] = eb_o[i];
            }
        }
        pending = 0;
    };
    // a 16 x 16 block with at least one score at or above its query's cut (rare: the cut is the KL-th best of the
    // rows already scanned).  Validity, order and the tie rule (at the cut's score: only orders <= the cut's) here.
    auto emit_block = [&](const f32x4 s, float cut, int qj, int64_t p0) {
#pragma unroll 1
        for (int e = 0; e < 4; ++e) {
            const int64_t p = p0 + e;
            const float sc = s[e];
            int64_t o64 = p - rv.head;
            if (o64 < 0) o64 += rv.cap;
            const int o = (int)o64;
            bool pass = p < r_hi && !(p >= dense.d0 && p < dense.d1) && sc >= cut;
            if (__ballot(pass && sc == cut)) {  // a tie with the cut itself
                const int to = qj < g.Q ? g.thr_o[qj] : -1;
                if (sc == cut && o > to) pass = false;
            }
            const unsigned long long m = __ballot(pass);
            if (m) {  // wave-uniform
                if (pending > GS_WBUF - 64) flush();
                if (pass) {
                    const int idx =
                        pending + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                    lds_wr32(eb0 + idx * 4, __builtin_bit_cast(unsigned, sc));
                    lds_wr32(eb0 + (GS_WBUF + idx) * 4, (unsigned)o);
                    lds_wr32(eb0 + (2 * GS_WBUF + idx) * 4, (unsigned)qj);
                }
                pending += __popcll(m);
            }
        }
    };
    auto epilogue = [&](int64_t row0, int q0, int par) {
        const unsigned tab = lds0 + GS_STAGE + par * GS_TAB;
        const unsigned rnb = tab + (wr * 128 + 4 * h) * 4;       // + 64 i
        const unsigned ctb = tab + 1024 + (wc * 64 + r16) * 4;   // + 64 j
        u32x4 rn[8];
        unsigned ct[4];
        // the twelve reads and their wait in ONE asm statement: the compiler's waitcnt insertion does not see asm-issued
        // LDS reads, so nothing - no copy, no spill of a destination - may come between the issue and the wait
        asm volatile(
            "ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:64\n\tds_read_b128 %2, %12 offset:128\n\t"
            "ds_read_b128 %3, %12 offset:192\n\tds_read_b128 %4, %12 offset:256\n\tds_read_b128 %5, %12 offset:320\n\t"
            "ds_read_b128 %6, %12 offset:384\n\tds_read_b128 %7, %12 offset:448\n\t"
            "ds_read_b32 %8, %13\n\tds_read_b32 %9, %13 offset:64\n\tds_read_b32 %10, %13 offset:128\n\t"
            "ds_read_b32 %11, %13 offset:192\n\ts_waitcnt lgkmcnt(0)"
            : "=&v"(rn[0]), "=&v"(rn[1]), "=&v"(rn[2]), "=&v"(rn[3]), "=&v"(rn[4]), "=&v"(rn[5]), "=&v"(rn[6]),
              "=&v"(rn[7]), "=&v"(ct[0]), "=&v"(ct[1]), "=&v"(ct[2]), "=&v"(ct[3])
            : "v"(rnb), "v"(ctb)
            : "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int qj = q0 + wc * 64 + 16 * j + r16;
            const float cut = qj < g.Q ? __builtin_bit_cast(float, ct[j]) : INFINITY;  // padded queries never emit
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 s = acc[i][j] * __builtin_bit_cast(f32x4, rn[i]);
                const float best = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
                if (__ballot(best >= cut) == 0ull) continue;
                emit_block(s, cut, qj, row0 + wr * 128 + 16 * i + 4 * h);
            }
        }
    };

    // ---- prologue ----
    coords(tile, cur_row0, cur_q0);
    set_sources(cur_row0, cur_q0);
    int par = 0;
    stage_tab(cur_row0, cur_q0, par);
    stage_Aa0(0, 0);
    stage_Bb0(0, 0);
    stage_Bb1(0, 0);
    stage_Aa1(0, 0);
    wait_vmcnt<0>();
    GS_BAR();
    if (wr == 1) GS_BAR();  // wave row 1 runs one barrier behind wave row 0

    // The K-tile stream of gemm256p_kernel: LDS-DMA of K-tile kt + 1 spread over the four phases of K-tile kt, issue
    // order Aa0, Bb0, Bb1, Aa1, each region retired by a counted wait before the first barrier of the phase that
    // precedes its first read.  At a tile boundary the next tile's table and its K-tile 1 are staged BEFORE the
    // epilogue (K-tile 0 went out during the last K-tile), so the first K-tile after a boundary has nothing to stage
    // and skips over those 9 operations in its counts.  The epilogue may leave stores behind (a flush): the counts
    // then also wait for them, which is always safe.
    int gk = 0;
    bool prestaged = false;
    while (true) {
        const int next_tile = tile + ncu;
        const bool has_next = next_tile < nt_x;
        if (has_next) coords(next_tile, nxt_row0, nxt_q0);
        for (int kt = 0; kt < nk; ++kt, ++gk) {
            const char *buf = smem + (gk & 1) * 4 * GS_HALF;
            const int nb = (gk + 1) & 1;
            int skt = kt + 1;
            bool more = skt < nk;
            if (!more && has_next) {  // roll over to the next tile's first K-tile
                set_sources(nxt_row0, nxt_q0);
                skt = 0;
                more = true;
            }
            const bool k0_after = prestaged && kt == 0;
            // phase 1
            read_a(buf, 0);
            read_b(buf, 0);
            if (k0_after) {
                wait_vmcnt<11>();  // retire Bb1 of this K-tile; younger: its Aa1 (2) + table (1) + K-tile 1 (8)
            } else if (more) {
                stage_Aa0(skt, nb);
                wait_vmcnt<4>();
            } else {
                wait_vmcnt<0>();
            }
            GS_BAR();
            GS_LGKM0();
            mma(0, 0, kt == 0);
            GS_BAR();
            // phase 2
            read_b(buf, 1);
            if (k0_after) {
                wait_vmcnt<9>();   // retire Aa1 of this K-tile; younger: table (1) + K-tile 1 (8)
            } else if (more) {
                stage_Bb0(skt, nb);
                wait_vmcnt<4>();
            }
            GS_BAR();
            GS_LGKM0();
            mma(0, 1, kt == 0);
            GS_BAR();
            // phase 3
            read_a(buf, 1);
            if (more && !k0_after) stage_Bb1(skt, nb);
            GS_BAR();
            GS_LGKM0();
            mma(1, 1, kt == 0);
            GS_BAR();
            // phase 4
            if (k0_after) {
                wait_vmcnt<4>();   // retire the table and Aa0 + Bb0 of K-tile 1; younger: its Bb1, Aa1 (4)
            } else if (more) {
                stage_Aa1(skt, nb);
                wait_vmcnt<4>();
            }
            GS_BAR();
            mma(1, 0, kt == 0);
            GS_BAR();
        }
        prestaged = false;
        // Tile boundary: wave row 0 waits one barrier so that both wave rows run their epilogues in the same interval;
        // wave row 1 takes its matching extra barrier after the epilogue (restores the one-barrier offset).
        if (wr == 0) GS_BAR();
        if (has_next) {
            const int cb = (gk + 1) & 1;  // == buffer of the K-tile just consumed
            stage_tab(nxt_row0, nxt_q0, par ^ 1);
            stage_Aa0(1, cb);
            stage_Bb0(1, cb);
            stage_Bb1(1, cb);
            stage_Aa1(1, cb);
        }
        epilogue(cur_row0, cur_q0, par);
        if (wr == 1) GS_BAR();
        if (!has_next) break;
        prestaged = true;
        tile = next_tile;
        cur_row0 = nxt_row0;
        cur_q0 = nxt_q0;
        par ^= 1;
    }
    if (wr == 0) GS_BAR();  // match wave row 1's extra barrier
    if (pending) flush();
#undef GS_BAR
#undef GS_LGKM0
}

}  // namespace

bool vm_topk_gscan_supported(const vm_memory *m, int Q, int64_t rows) {
    // from 129 queries on: up to 128 the emit scan's single superblock is HBM-bound and as fast; 160 / 384 queries
    // over 1 M x 768 rows: 0.47 -> 0.44 ms / 0.92 -> 0.76 ms (the emit scan needs a second, half-empty superblock)
    static const int min_q = (int)VM_DEV_ENV("GSCAN_MINQ", 129);
    if (min_q <= 0 || Q < min_q || rows < 16384) return false;
    // enough 256 x 256 tiles for every CU to walk a few (a persistent tile walk with fewer leaves CUs idle for most of
    // the launch: the emit scan's 32-row tiles spread such a range better)
    const int64_t tiles = ((rows + 255) / 256) * ((Q + 255) / 256);
    return tiles >= 4 * (int64_t)m->ctx->num_cus && m->D % 128 == 0 && m->D <= 2048 && m->ctx->num_cus % 8 == 0;
}

// Candidates of rows [row_begin, min(n, row_limit)) at or above each query's cut are appended to the candidate buffers
// (cand_cnt is NOT reset here).  row_begin must be a multiple of 256.
int vm_topk_gscan(vm_memory *m, const void *queries, int Q, int q_thr, const float *thr_s, const int *thr_o,
                  int *cand_cnt, float *cand_s, int *cand_o, int64_t row_begin, int64_t row_limit, hipStream_t st) {
    if (row_begin % 256 != 0 || !thr_s || !thr_o)
        return vm_fail(m->ctx, VM_ERR_INVALID, "gscan: row_begin %lld must be a multiple of 256 and cuts are required",
                       (long long)row_begin);
    GscanArgs g;
    g.mem = m->rows;
    g.rnorm = m->rnorm32;
    g.queries = (const uint16_t *)queries;
    g.d_total = m->d_total;
    g.cap = m->cap;
    g.cap_pad = (m->cap + 63) / 64 * 64;
    g.ring = m->ring;
    g.D = m->D;
    g.Q = Q;
    g.q_thr = q_thr;
    g.thr_s = thr_s;
    g.thr_o = thr_o;
    g.cand_cnt = cand_cnt;
    g.cand_s = cand_s;
    g.cand_o = cand_o;
    g.row_begin = row_begin;
    g.row_limit = row_limit;
    // XCD grid: the smallest query split that keeps an XCD's query tiles within its L2 (8 tiles = 3 MiB at D = 768)
    // among those with the fewest tiles on the busiest XCD
    const int nqt = (Q + 255) / 256;
    const int64_t rows = (m->cap < row_limit ? m->cap : row_limit) - row_begin;
    const int64_t nrp = rows > 0 ? (rows + 255) / 256 : 1;
    const int fit = (int)(((size_t)3 << 20) / ((size_t)256 * m->D * 2));  // query tiles an L2 keeps beside the row stream
    int best_qx = 8;
    int64_t best_cost = -1;
    for (int qx = 8; qx >= 1; qx >>= 1) {
        const int per = (nqt + qx - 1) / qx;
        const int rxn = 8 / qx;
        int64_t cost = (int64_t)per * ((nrp + rxn - 1) / rxn);
        if (per > (fit > 1 ? fit : 1)) cost = cost * 3 / 2;  // spills L2: every tile re-fetches its queries from MALL
        if (best_cost < 0 || cost <= best_cost) {
            best_cost = cost;
            best_qx = qx;
        }
    }
    static const int qx_env = (int)VM_DEV_ENV("GSCAN_QX", 0);
    g.QX = (qx_env == 1 || qx_env == 2 || qx_env == 4 || qx_env == 8) ? qx_env : best_qx;
    auto kern = m->dtype == VM_F16 ? topk_gscan_kernel<VM_F16> : topk_gscan_kernel<VM_BF16>;
    static unsigned long long attr_set[2] = {0, 0};   // per dtype, one bit per device
    if (!((attr_set[m->dtype == VM_F16 ? 0 : 1] >> (m->ctx->device & 63)) & 1ull)) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, GS_LDS);
        if (e != hipSuccess) return vm_fail(m->ctx, VM_ERR_HIP, "gscan LDS opt-in %d: %s", GS_LDS, hipGetErrorString(e));
        attr_set[m->dtype == VM_F16 ? 0 : 1] |= 1ull << (m->ctx->device & 63);
    }
    vm_prof_scope prof(m->ctx, VM_PROF_TOPK_SCAN, st);
    kern<<<m->ctx->num_cus, 512, GS_LDS, st>>>(g);
    VM_LAUNCH_CHECK(m->ctx);
    return VM_OK;
}
