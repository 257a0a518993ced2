#!/bin/bash
# round 4, GPU call 6: intra-XCD ramp de-phasing of the persistent GEMM; attention after the revert
set -o pipefail
O=gpurun_out/r4f
mkdir -p $O
B=tools/bin
run() { VIDMEM_GEMM_DEPHASE_RAMP=$1 $B/gemm_bench $2 $3 $4 $5 20 2>&1 | grep "persistent" | sed "s/^/ramp $1: /; s/maxerr.*//" | tee -a $O/ramp.log; }
for r in 1 2; do for p in 0 2 5 10 20 40; do run $p 173360 3072 768 1; done; done
for p in 0 5 10 20 0 5 10 20; do run $p 173360 2304 768 0; done
for p in 0 5 10 20 0 5 10 20; do run $p 173360 768 768 0; done
for p in 0 5 10 20 0 5 10 20; do run $p 173360 768 3072 0; done
for i in 1 2; do python tools/enc_probe.py clip_l14_336 bf16 224 2 3 2>&1 | grep "attention" | sed "s/.*'attention': (\([0-9.]*\), \([0-9]*\)).*/attention \1 ms in \2 launches/" | tee -a $O/attn.log; done
python -m pytest tests/test_encoder_gpu.py -q -k "short_stacks or full_models or bench_size" 2>&1 | tail -3
exit 0
