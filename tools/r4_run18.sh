#!/bin/bash
O=gpurun_out/r4s; mkdir -p $O
for c in "vit_b16_224 f16" "vit_b16_224_heavy f16" "clip_l14_336 bf16" "clip_l14_336_heavy bf16"; do python tools/golden_probe.py $c 2>&1 | grep "rep 0" | tee -a $O/golden.log; done
exit 0
