#!/bin/bash
# round 4, GPU call 13: the persistent 577-token attention (next item's K / V rows prefetched into idle waves' registers):
# parity tests, then alternating A/B against the one-item kernel (developer library), then the pair kernel's compile-time
# ablation table (one-item kernel)
set -o pipefail
O=gpurun_out/r4n; mkdir -p $O
python -m pytest tests/test_encoder_gpu.py -q -x -k "long_attention or short_stacks or last_layer or bench_size or full_models or two_stream" > $O/enc_tests.log 2>&1; trc=$?
tail -5 $O/enc_tests.log
[ $trc -ne 0 ] && exit $trc
at() { TOOLS_DEV=1 VIDMEM_ATTN_PERSIST=$1 VIDMEM_ATTN_ABL=$2 python tools/enc_probe.py clip_l14_336 bf16 224 2 2>&1 | grep -E "attention|frames/s" | sed "s/.*'attention': (\([0-9.]*\), \([0-9]*\)).*/attention \1 ms in \2 launches/" | sed "s/^/persist $1 abl $2: /" | tee -a $O/attn.log; }
for r in 1 2; do at 0 0; at 1 0; done
for a in 1 2 4 8 16 32 3 10 48 63; do at 0 $a; done
exit 0
