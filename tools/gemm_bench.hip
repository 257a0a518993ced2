// Developer harness (not part of libvidmem): correctness + timing of the encoder GEMM kernels on random data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_bench.hip <pkg>/csrc/gemm.o <pkg>/csrc/context.o -o gpurun_out/gemm_bench
//   gemm_bench M N K [epi] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
#include "../real-time-brain-inspired-video-memory_amd/csrc/vm_kernels.h"
void vm_gemm_set_variant(int v);

__global__ void ref_gemm(const _Float16 *X, const _Float16 *W, const float *bias, float *C, int M, int N, int K) {
    int f = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (f >= N) return;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += (float)X[(size_t)t * K + k] * (float)W[(size_t)f * K + k];
    C[(size_t)t * N + f] = acc + bias[f];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

int main(int argc, char **argv) {
    int M = argc > 1 ? atoi(argv[1]) : 21670, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
    int epi = argc > 4 ? atoi(argv[4]) : EPI_STORE16, iters = argc > 5 ? atoi(argv[5]) : 20;
    vm_ctx *ctx; if (vm_init(0, &ctx)) { printf("vm_init failed: %s\n", vm_last_error(nullptr)); return 1; }
    std::vector<_Float16> hx((size_t)M * K), hw((size_t)N * K); std::vector<float> hb(N);
    srand(1);
    const bool zero = getenv("ZERO") != nullptr;   // all-zero operands: same instruction stream, minimal switching power
    for (auto &v : hx) v = zero ? (_Float16)0.f : (_Float16)(rand() / (float)RAND_MAX * 2.f - 1.f);
    for (auto &v : hw) v = zero ? (_Float16)0.f : (_Float16)((rand() / (float)RAND_MAX * 2.f - 1.f) * 0.05f);
    for (auto &v : hb) v = rand() / (float)RAND_MAX - 0.5f;
    _Float16 *dx, *dw, *dout16; float *db, *dref, *dout32;
    CK(hipMalloc(&dx, hx.size() * 2)); CK(hipMalloc(&dw, hw.size() * 2)); CK(hipMalloc(&db, N * 4));
    CK(hipMalloc(&dref, (size_t)M * N * 4)); CK(hipMalloc(&dout16, (size_t)M * N * 2)); CK(hipMalloc(&dout32, (size_t)M * N * 4));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
    ref_gemm<<<dim3((N + 255) / 256, M), 256>>>(dx, dw, db, dref, M, N, K);
    CK(hipDeviceSynchronize());
    std::vector<float> href((size_t)M * N); CK(hipMemcpy(href.data(), dref, href.size() * 4, hipMemcpyDeviceToHost));
    GemmArgs g{}; g.prof_cat = VM_PROF_GEMM_QKV; g.X = (const uint16_t *)dx; g.W = (const uint16_t *)dw; g.bias = db; g.out16 = (uint16_t *)dout16; g.out32 = dout32;
    g.M = M; g.N = N; g.K = K; g.ldx = K; g.ldo = N;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int vlist[] = {3, 1024 + 256, 1024 + 32, 1024 + 64, 1024 + 96, 1024 + 112, 1024 + 16, 1024 + 8, 1, 2, 2 + 128, 2 + 16, 2 + 32, 2 + 48, 2 + 64, 2 + 80, 2 + 112};
    int nv = getenv("ABLATE") ? 8 : (getenv("ALLV") ? 11 : 1);
    std::vector<uint16_t> base16;  // raw 16-bit output of the first variant: later variants must match it bit for bit
    if (const char *vs = getenv("VARIANTS")) {  // comma-separated list, run in that order
        nv = 0;
        for (const char *q = vs; *q && nv < 13;) { vlist[nv++] = atoi(q); while (*q && *q != ',') ++q; if (*q) ++q; }
    }
    for (int vi = 0; vi < nv; ++vi) {
        int variant = vlist[vi];
        if (variant >= 2 && N % 256) continue;
        vm_gemm_set_variant(variant);
        CK(hipMemset(dout16, 0, (size_t)M * N * 2)); CK(hipMemset(dout32, 0, (size_t)M * N * 4));
        if (vm_gemm(ctx, VM_F16, g, epi, 0)) { printf("vm_gemm: %s\n", vm_last_error(ctx)); return 1; }
        CK(hipDeviceSynchronize());
        double maxerr = 0, maxref = 0; size_t bad = 0;
        if (epi == EPI_GELU16 || epi == EPI_QGELU16) {   // activation of the fp32 reference in double, against the fp16 output
            std::vector<_Float16> ho((size_t)M * N); CK(hipMemcpy(ho.data(), dout16, ho.size() * 2, hipMemcpyDeviceToHost));
            double maxfn = 0;   // largest deviation of the ACTIVATION itself: output - round16(exact activation of the device's own pre-activation) is not observable; use the fp32 reference's
            for (size_t i = 0; i < ho.size(); ++i) {
                const double x = href[i];
                const double want = epi == EPI_GELU16 ? 0.5 * x * erfc(-x * 0.70710678118654752440) : x / (1.0 + exp(-1.702 * x));
                const double d = fabs((double)ho[i] - want);
                if (d > maxerr) maxerr = d;
                if (fabs(want) > maxref) maxref = fabs(want);
                // fp16 rounding of the output (2^-11 relative) + the fp32 accumulation-order difference of the pre-activation
                if (d > 6e-4 * fabs(want) + 3e-5) { ++bad; if (d - 6e-4 * fabs(want) > maxfn) maxfn = d - 6e-4 * fabs(want); }
            }
            if (bad) printf("  activation check: %zu elements beyond 6e-4 |y| + 3e-5, worst excess %.3g\n", bad, maxfn);
        } else if (epi == EPI_STORE16) {
            std::vector<_Float16> ho((size_t)M * N); CK(hipMemcpy(ho.data(), dout16, ho.size() * 2, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < ho.size(); ++i) { double d = fabs((double)ho[i] - href[i]); if (d > maxerr) maxerr = d; if (fabs(href[i]) > maxref) maxref = fabs(href[i]); if (d > 2e-3 * (1 + fabs(href[i]))) ++bad; }
        } else {
            std::vector<float> ho((size_t)M * N); CK(hipMemcpy(ho.data(), dout32, ho.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < ho.size(); ++i) { double d = fabs((double)ho[i] - href[i]); if (d > maxerr) maxerr = d; if (fabs(href[i]) > maxref) maxref = fabs(href[i]); if (d > 2e-4 * (1 + fabs(href[i]))) ++bad; }
        }
        size_t diff16 = 0;
        if (epi != EPI_RESID32 && epi != EPI_PATCH) {
            std::vector<uint16_t> raw((size_t)M * N); CK(hipMemcpy(raw.data(), dout16, raw.size() * 2, hipMemcpyDeviceToHost));
            if (base16.empty()) base16 = raw;
            else for (size_t i = 0; i < raw.size(); ++i) diff16 += raw[i] != base16[i];
        }
        for (int i = 0; i < 3; ++i) vm_gemm(ctx, VM_F16, g, epi, 0);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) vm_gemm(ctx, VM_F16, g, epi, 0);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
        if (getenv("STAMPS") && variant == 3) {   // tile-boundary wall-clock stamps of one more launch: how far apart do the CUs run?
            unsigned long long *ds; CK(hipMalloc(&ds, 256 * 64 * 2 * 8)); CK(hipMemset(ds, 0, 256 * 64 * 2 * 8));
            GemmArgs gs = g; gs.stamps = ds;
            vm_gemm(ctx, VM_F16, gs, epi, 0); CK(hipDeviceSynchronize());
            std::vector<unsigned long long> hs(256 * 64 * 2); CK(hipMemcpy(hs.data(), ds, hs.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull; for (int b = 0; b < 256; ++b) if (hs[b * 128] && hs[b * 128] < t0) t0 = hs[b * 128];
            // per round: spread of the "K loop done" stamps over the CUs (min / median / max, us after the first CU's first
            // boundary), the same within XCD 0 (blockIdx % 8 == 0), and the median epilogue duration (stamp 1 - stamp 0)
            for (int r = 0; r < 64; r += (r < 4 ? 1 : 4)) {
                std::vector<double> a, x0, ep;
                for (int b = 0; b < 256; ++b) { const unsigned long long s0 = hs[(b * 64 + r) * 2], s1 = hs[(b * 64 + r) * 2 + 1]; if (!s0 || !s1) continue; a.push_back((s0 - t0) / 100.0); ep.push_back((s1 - s0) / 100.0); if (b % 8 == 0) x0.push_back((s0 - t0) / 100.0); }
                if (a.size() < 8) break;
                std::sort(a.begin(), a.end()); std::sort(x0.begin(), x0.end()); std::sort(ep.begin(), ep.end());
                printf("  round %2d: K-loop-done over %3zu CUs: min %7.2f median %7.2f max %7.2f us (spread %.2f; XCD 0: %.2f); epilogue median %.2f max %.2f us\n", r, a.size(), a.front(), a[a.size() / 2], a.back(), a.back() - a.front(), x0.empty() ? 0.0 : x0.back() - x0.front(), ep[ep.size() / 2], ep.back());
            }
            CK(hipFree(ds));
        }
        printf("M=%d N=%d K=%d epi=%d variant=%s: %.1f us  %.0f TFLOP/s  maxerr %.3g (ref max %.3g) bad=%zu bits-differ=%zu\n", M, N, K, epi,
               variant == 1280 ? "256p default-policy stores" : variant == 1056 ? "256p X-panel0" : variant == 1088 ? "256p W-tile0" : variant == 1120 ? "256p X0+W0" : variant == 1136 ? "256p X0+W0 noEPI" : variant == 1 ? "128^2" : (variant == 2 ? "256^2" : variant == 3 ? "256^2 persistent" : (variant == 1032 ? "256p noSTORE" : variant == 1040 ? "256p noEPILOGUE" : variant == 130 ? "256 noSTORE" : variant == 18 ? "256 noDMA" : variant == 34 ? "256 noDSREAD" : variant == 50 ? "256 noDMA noDSREAD" : variant == 66 ? "256 noMFMA" : variant == 82 ? "256 noDMA noMFMA" : "256 none")), ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, maxerr, maxref, bad, diff16);
    }
    return 0;
}
