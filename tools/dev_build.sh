#!/bin/bash
# Developer helper: rebuild libvidmem.so and the GEMM harness (absolute paths; safe from any cwd).
# The harness links its own gemm object compiled with -DVM_GEMM_ABLATE (the ablation variants are not in libvidmem.so).
set -e
R=/root/repo
P=$R/real-time-brain-inspired-video-memory_amd
make -s -j8 -C $P/csrc 2>&1 | grep -E "error|warning: unused|Error" -A5 || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c $R/tools/gemm_bench.hip -o /tmp/gemm_bench.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DVM_GEMM_ABLATE -I$R/include -c $P/csrc/gemm.hip -o /tmp/gemm_abl.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/gemm_bench.o /tmp/gemm_abl.o $P/csrc/context.o -o $R/tools/bin/gemm_bench
ls -la $R/tools/bin/gemm_bench $P/libvidmem.so
