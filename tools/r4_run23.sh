#!/bin/bash
# round 4, GPU call 23: three-image streaming attention (197 tokens; loader two items ahead, queries in registers):
# parity tests, then alternating A/B against the two-image kernel (developer library)
set -o pipefail
O=gpurun_out/r4x; mkdir -p $O
python -m pytest tests/test_encoder_gpu.py tests/test_dropin_gpu.py tests/test_streaming_gpu.py -q -x > $O/tests.log 2>&1; trc=$?
tail -4 $O/tests.log
[ $trc -ne 0 ] && exit $trc
for r in 1 2 3; do for v in 0 1; do
  TOOLS_DEV=1 VIDMEM_ATTN_STREAM3=$v python tools/enc_probe.py vit_b16_224 f16 880 3 2>&1 | grep -E "attention|frames/s" | sed "s/.*'attention': (\([0-9.]*\), \([0-9]*\)).*/attention \1 ms in \2 launches/" | sed "s/^/stream3 $v: /" | tee -a $O/ab.log
done; done
exit 0
