#!/bin/bash
# Developer helper: rebuild libvidmem.so, the developer library (VIDMEM_* switches) and the GEMM harness
# (absolute paths; safe from any cwd).  The harness links its own gemm object compiled with -DVM_GEMM_ABLATE (the ablation
# variants are not in libvidmem.so); gemm_bench_poly is the same with the round-2 polynomial GELU (-DVM_GELU_POLY=1).
set -e
R=/root/repo
P=$R/real-time-brain-inspired-video-memory_amd
make -s -j8 -C $P/csrc all dev 2>&1 | grep -E "error|warning: unused|Error" -A5 || true
mkdir -p $R/tools/bin
HC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17"
$HC -c $R/tools/gemm_bench.hip -o /tmp/gemm_bench.o
$HC -fPIC -ffp-contract=off -DVM_GEMM_ABLATE -DVM_DEV_SWITCHES -I$R/include -c $P/csrc/gemm.hip -o /tmp/gemm_abl.o &
$HC -fPIC -ffp-contract=off -DVM_GEMM_ABLATE -DVM_DEV_SWITCHES -DVM_GELU_POLY=1 -I$R/include -c $P/csrc/gemm.hip -o /tmp/gemm_abl_poly.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/gemm_bench.o /tmp/gemm_abl.o $P/csrc/context.o -o $R/tools/bin/gemm_bench
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/gemm_bench.o /tmp/gemm_abl_poly.o $P/csrc/context.o -o $R/tools/bin/gemm_bench_poly
ls -la $R/tools/bin/gemm_bench $R/tools/bin/gemm_bench_poly $P/libvidmem.so $P/libvidmem_dev.so
