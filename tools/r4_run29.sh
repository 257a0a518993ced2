#!/bin/bash
# round 4, GPU call 29: on two streams, the low-register LayerNorm against the ordinary one (developer switch), alternating
O=gpurun_out/r4E; mkdir -p $O
for r in 1 2 3; do for v in 1 0; do
  TOOLS_DEV=1 VIDMEM_LN_LOWREG=$v python tools/enc_probe.py vit_b16_224 f16 1760 6 2>&1 | grep "frames/s" | sed "s/^/ln_lowreg $v: /" | tee -a $O/ab.log
done; done
for r in 1 2; do for v in 1 0; do
  TOOLS_DEV=1 VIDMEM_LN_LOWREG=$v python tools/enc_probe.py clip_l14_336 bf16 448 3 2>&1 | grep "frames/s" | sed "s/^/ln_lowreg $v: /" | tee -a $O/ab.log
done; done
exit 0
