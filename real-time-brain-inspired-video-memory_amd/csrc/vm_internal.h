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
