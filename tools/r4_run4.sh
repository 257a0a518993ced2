#!/bin/bash
# round 4, GPU call 4: XCD de-phasing of the persistent GEMM (start delay per XCD slot, percent of a tile time / 8)
set -o pipefail
O=gpurun_out/r4d
mkdir -p $O
B=tools/bin
run() { VIDMEM_GEMM_DEPHASE_PCT=$1 $B/gemm_bench $2 $3 $4 $5 20 2>&1 | grep "persistent" | sed "s/^/dephase $1%: /" | tee -a $O/dephase.log; }
for r in 1 2; do
  for p in 0 50 100 200 400; do run $p 173360 3072 768 1; done
done
for p in 0 100 200 0 100 200; do run $p 173360 2304 768 0; done
for p in 0 100 200 0 100 200; do run $p 173360 768 768 0; done
for p in 0 100 200 0 100 200; do run $p 173360 768 3072 0; done
# CLIP-L shapes (224 frames x 577 tokens = 129,248 rows)
for p in 0 100 200 0 100 200; do run $p 129248 4096 1024 2; done
for p in 0 100 200 0 100 200; do run $p 129248 1024 4096 0; done
exit 0
