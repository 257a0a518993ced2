// Developer probe (not part of libvidmem): what does a CU's store path sustain on the GEMM epilogue's store pattern?
// One 512-thread workgroup per CU (128 KiB of LDS claimed); every wave issues ROUNDS x 16 global_store_dwordx4 of
// 4 rows x 256 B (the persistent GEMM's epilogue: 128 KiB per CU and tile), timed with s_memtime inside the kernel.
//   store_probe [active_every=1] [nt=1] [rounds=8] [ldo_elems=2304] [xcd_only=-1] [pattern=0]   (pattern 0: 4 rows x 256 B per store, 1: 16 rows x 64 B, 2: 8 rows x 128 B; xcd_only = x: only the CUs of XCD x store)
// active_every = n: only every n-th CU slot of each XCD stores (blockIdx / 8 % n == 0), the others exit at once:
// 1 = all 32 CUs of an XCD burst together, 32 = one CU per XCD.  Build: hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o tools/bin/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ void __launch_bounds__(512, 1) store_kernel(unsigned short *out, int ldo, int rounds, int every, long long *cyc, int panels, int xcd_only, int pattern) {
    extern __shared__ char smem[];
    const int slot = blockIdx.x >> 3;
    if (slot % every) return;
    if (xcd_only >= 0 && (int)(blockIdx.x & 7) != xcd_only) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 2, wc = wave & 3, row0 = lane >> 4, ch = lane & 15;
    if (threadIdx.x == 0) smem[0] = 1;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
    for (int r = 0; r < rounds; ++r) {
        const int tile = (blockIdx.x + r * gridDim.x) % panels;     // a different 256-row panel every round
        const size_t t_base = (size_t)tile * 256 + wc * 64;
        if (pattern == 1) {   // 16 rows x 64 B per instruction (the lane-row-swap epilogue): lane (r16, h) -> row r16, 16 B piece
            const int r16 = lane & 15, h = lane >> 4;
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int ii = 0; ii < 8; ii += 2) {
                    unsigned short *d = out + (t_base + 16 * p + r16) * ldo + wr * 128 + 16 * (ii + (h & 1)) + 8 * (h >> 1);
                    if (NT) __builtin_nontemporal_store(v, (u32x4 *)d); else *(u32x4 *)d = v;
                }
            continue;
        }
        if (pattern == 2) {   // 8 rows x 128 B per instruction: lane (r16, h) -> row r16 & 7, piece (r16 >> 3) * 4 + slot(h)
            const int r16 = lane & 15, h = lane >> 4;
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int grp = 0; grp < 2; ++grp)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        unsigned short *d = out + (t_base + 16 * p + 8 * u + (r16 & 7)) * ldo + wr * 128 + 64 * grp + ((r16 & 8) << 2) + 16 * (h & 1) + 8 * (h >> 1);
                        if (NT) __builtin_nontemporal_store(v, (u32x4 *)d); else *(u32x4 *)d = v;
                    }
            continue;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const size_t t = t_base + 16 * p + 8 * half + row0;
                unsigned short *d0 = out + t * ldo + wr * 128 + ch * 8, *d1 = d0 + (size_t)4 * ldo;
                if (NT == 1) { __builtin_nontemporal_store(v, (u32x4 *)d0); __builtin_nontemporal_store(v, (u32x4 *)d1); }
                else if (NT == 2) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(d0), "v"(v) : "memory"); asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(d1), "v"(v) : "memory"); }
                else if (NT == 3) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(d0), "v"(v) : "memory"); asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(d1), "v"(v) : "memory"); }
                else if (NT == 4) { asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(d0), "v"(v) : "memory"); asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(d1), "v"(v) : "memory"); }
                else { *(u32x4 *)d0 = v; *(u32x4 *)d1 = v; }
            }
    }
    const long long t1 = __builtin_readcyclecounter();   // all stores ISSUED
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = __builtin_readcyclecounter();   // all stores acknowledged
    __syncthreads();
    if (lane == 0) { cyc[(blockIdx.x * 8 + wave) * 2] = t1 - t0; cyc[(blockIdx.x * 8 + wave) * 2 + 1] = t2 - t0; }
}

int main(int argc, char **argv) {
    const int every = argc > 1 ? atoi(argv[1]) : 1, nt = argc > 2 ? atoi(argv[2]) : 1, rounds = argc > 3 ? atoi(argv[3]) : 8;
    const int ldo = argc > 4 ? atoi(argv[4]) : 2304, panels = 678, xcd_only = argc > 5 ? atoi(argv[5]) : -1, pattern = argc > 6 ? atoi(argv[6]) : 0;
    unsigned short *out; long long *cyc;
    CK(hipMalloc(&out, (size_t)panels * 256 * ldo * 2)); CK(hipMalloc(&cyc, 256 * 8 * 2 * 8)); CK(hipMemset(cyc, 0, 256 * 8 * 2 * 8));
    auto k = nt == 1 ? store_kernel<1> : nt == 2 ? store_kernel<2> : nt == 3 ? store_kernel<3> : nt == 4 ? store_kernel<4> : store_kernel<0>;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0, 0));
        k<<<256, 512, 131072, 0>>>(out, ldo, rounds, every, cyc, panels, xcd_only, pattern);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(256 * 8 * 2); CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> iss, ack;
    for (int b = 0; b < 256; ++b) { if ((b >> 3) % every) continue; if (xcd_only >= 0 && (b & 7) != xcd_only) continue; long long mi = 0, ma = 0; for (int w = 0; w < 8; ++w) { mi = std::max(mi, h[(b * 8 + w) * 2]); ma = std::max(ma, h[(b * 8 + w) * 2 + 1]); } iss.push_back((double)mi / rounds); ack.push_back((double)ma / rounds); }
    std::sort(iss.begin(), iss.end()); std::sort(ack.begin(), ack.end());
    const double kb = 128.0;
    printf("pattern %d: active CUs %zu (every %d), %s stores, %d x 128 KiB per CU, ldo %d: kernel %.1f us; s_memtime ticks per 128 KiB: issue median %.0f (max %.0f), acknowledged median %.0f (max %.0f) -> %.2f TB/s chip by the events (ticks per us of the kernel: %.0f)\n",
           pattern, iss.size(), every, nt == 1 ? "nt" : nt == 2 ? "sc1" : nt == 3 ? "sc0 sc1" : nt == 4 ? "sc1 nt" : "plain", rounds, ldo, ms * 1e3, iss[iss.size() / 2], iss.back(), ack[ack.size() / 2], ack.back(),
           iss.size() * kb * 1024 * rounds / (ms * 1e-3) / 1e12, ack.back() * rounds / (ms * 1e3));
    return 0;
}
