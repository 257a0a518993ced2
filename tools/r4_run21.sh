#!/bin/bash
# round 4, GPU call 21: start offset of the second encoder stream (developer switch VIDMEM_ENC_PHASE), 1,760 frames = two passes
O=gpurun_out/r4v; mkdir -p $O
for r in 1 2; do for ph in 0 1 2 3 4 5; do
  TOOLS_DEV=1 VIDMEM_ENC_PHASE=$ph python tools/enc_probe.py vit_b16_224 f16 1760 6 2>&1 | grep "frames/s" | sed "s/^/phase $ph: /" | tee -a $O/phase.log
done; done
exit 0
