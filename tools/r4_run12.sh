#!/bin/bash
# round 4, GPU call 12: cache policy of the GEMM's output stores (nt = product; sc1 / sc0 sc1 / sc1 nt / sc0 / plain):
# the store path alone, then the four encoder GEMM shapes, alternating
O=gpurun_out/r4m; mkdir -p $O
B=tools/bin
for nt in 1 2 3 4 0; do for e in 1 8; do timeout -k 5 60 $B/store_probe $e $nt 16 2304 | tee -a $O/store_policy.log; done; done
ab() { for r in 1 2; do for pol in 1 2 3 4 6; do VIDMEM_GEMM_STORE_POLICY=$pol timeout -k 5 120 $B/gemm_bench 173360 $1 $2 $3 20 2>&1 | grep -E "variant" | sed "s/maxerr.*//" | sed "s/^/policy $pol: /" | tee -a $O/ab.log; done; done; }
ab 2304 768 0
ab 3072 768 1
ab 768 768 0
ab 768 3072 0
exit 0
