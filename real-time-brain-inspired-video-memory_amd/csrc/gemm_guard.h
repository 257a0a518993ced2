// Address-range guard of the 256 x 256 GEMM kernels (gemm.hip), kept free of HIP headers so that a host-only unit
// test can compile it (tests/test_abi.py::test_gemm_tile_guard).
//
// gemm256p_kernel / gemm256_kernel stage a tile through buffer descriptors whose base is the tile's first row: every
// byte offset inside a tile - per-lane row and chunk (255 rows x row stride + one row of K), plus the K-tile in the
// wave-uniform soffset - must stay below 2^31 (num_records is a signed 32-bit field in the builtin, offsets are
// unsigned 32-bit).  Shapes beyond that take the 128 x 128 kernel, which uses 64-bit per-lane pointers.
#pragma once
#include <cstdint>

inline bool vm_gemm256_tile_addressable(int64_t K, int64_t ldx) {
    const int64_t lim = (int64_t)1 << 31;
    if (K <= 0 || ldx < K) return false;
    const int64_t w_tile = 256 * K * 2;               // bytes of a 256-row weight tile (rows are K apart)
    const int64_t x_tile = (255 * ldx + K) * 2;       // bytes from the first row of a token panel to the end of its last
    return w_tile < lim && x_tile < lim;
}
