#!/bin/bash
# round 4, GPU call 25: the GEMM at 224 VGPRs (LayerNorm of the other stream fits beside it) against the build before it
# (227-228), alternating on one box: ViT-B/16 1,760 frames (two passes, two streams) and CLIP-L 448 frames
O=gpurun_out/r4A; mkdir -p $O
for r in 1 2 3; do for l in prev head; do
  TOOLS_LIB=tools/bin/libvidmem_$l.so python tools/enc_probe.py vit_b16_224 f16 1760 6 2>&1 | grep "frames/s" | sed "s/^/$l: /" | tee -a $O/ab.log
done; done
for r in 1 2; do for l in prev head; do
  TOOLS_LIB=tools/bin/libvidmem_$l.so python tools/enc_probe.py clip_l14_336 bf16 448 3 2>&1 | grep "frames/s" | sed "s/^/$l: /" | tee -a $O/ab.log
done; done
exit 0
