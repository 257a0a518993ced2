#!/bin/bash
# round 4, GPU call 31: configs[2]'s top-20 over 1 M x 1024 bf16 (emit scan, two LDS stages at D = 1024): deep staging
# schedule for one query group per wave (developer switch bit 0) against the default, alternating
O=gpurun_out/r4G; mkdir -p $O
for r in 1 2 3; do for d in 2 3; do
  TOOLS_DEV=1 VIDMEM_EMIT_DEEP=$d python tools/knn_probe.py 1024 bf16 20 1000000 16 2>&1 | grep "Q=16" | sed "s/^/emit_deep $d: /" | tee -a $O/deep.log
done; done
for d in 2 3; do TOOLS_DEV=1 VIDMEM_EMIT_DEEP=$d python tools/knn_probe.py 768 f16 10 1000000 64 2>&1 | grep "Q=64" | sed "s/^/emit_deep $d: /" | tee -a $O/deep.log; done
exit 0
