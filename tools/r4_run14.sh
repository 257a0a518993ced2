#!/bin/bash
# round 4, GPU call 14: (a) the one-tile list scan (kNN Q = 16 over 1 M x 768) with 8 / 12 / 24 row loads per batch;
# (b) the 197-token attention with non-temporal context stores; alternating, developer library
O=gpurun_out/r4o; mkdir -p $O
for r in 1 2; do for lb in 8 12 24; do
  TOOLS_DEV=1 VIDMEM_SCAN_LB=$lb python tools/knn_probe.py 768 f16 10 1000000 16 2>&1 | grep "Q=16" | sed "s/^/scan_lb $lb: /" | tee -a $O/knn_lb.log
done; done
for r in 1 2 3; do for nt in 0 1; do
  TOOLS_DEV=1 VIDMEM_ATTN_CTX_NT=$nt python tools/enc_probe.py vit_b16_224 f16 880 3 2>&1 | grep -E "attention|frames/s" | sed "s/.*'attention': (\([0-9.]*\), \([0-9]*\)).*/attention \1 ms in \2 launches/" | sed "s/^/ctx_nt $nt: /" | tee -a $O/attn_nt.log
done; done
exit 0
