#!/usr/bin/env python3
"""Developer probe (not part of the product path): what the vendor BLAS reaches on the encoder's GEMM shapes, as a
yardstick for csrc/gemm.hip.  torch.nn.functional.linear dispatches to hipBLASLt / rocBLAS.  Same random operands as
tools/gemm_bench.hip (uniform [-1, 1) activations, 0.05-scaled weights), 16-bit in, 16-bit out, fp32 accumulate."""
import sys, torch
dev = "cuda:0"
torch.manual_seed(1)
shapes = [("qkv", 86877, 2304, 768), ("fc1", 86877, 3072, 768), ("fc2", 86877, 768, 3072), ("proj", 86877, 768, 768)]
for dt in (torch.float16, torch.bfloat16):
    for name, M, N, K in shapes:
        x = (torch.rand(M, K, device=dev) * 2 - 1).to(dt)
        w = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).to(dt)
        b = (torch.rand(N, device=dev) - 0.5).to(dt)
        for _ in range(3):
            y = torch.nn.functional.linear(x, w, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 10
        e0.record()
        for _ in range(it):
            y = torch.nn.functional.linear(x, w, b)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / it * 1e3
        print(f"{str(dt)[6:]:9s} {name:5s} M={M} N={N} K={K}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.0f} TFLOP/s", flush=True)
