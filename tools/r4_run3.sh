#!/bin/bash
# round 4, GPU call 3: where the FC1 epilogue's time goes with the table GELU (ablations), the emit route for KL >= 32
# (A/B through the developer library), the top-k suite, the uncertified-query probe, attention_pair ablations
set -o pipefail
O=gpurun_out/r4c
mkdir -p $O
B=tools/bin
VARIANTS=3,1032,1040,3 $B/gemm_bench 173360 3072 768 1 20 2>&1 | grep -v amdgpu.ids | sed "s/^/table: /" | tee $O/gelu_abl.log
VARIANTS=3,1032,1040,3 $B/gemm_bench_poly 173360 3072 768 1 20 2>&1 | grep -v amdgpu.ids | sed "s/^/poly:  /" | tee -a $O/gelu_abl.log
for e in 0 1 0 1; do
  TOOLS_DEV=1 VIDMEM_EMIT_KL32=$e python tools/knn_probe.py 1024 bf16 20 1000000 16,32 2>&1 | grep -v amdgpu.ids | sed "s/^/emit_kl32=$e: /" | tee -a $O/knn_kl32.log
done
TOOLS_DEV=1 VIDMEM_EMIT_KL32=0 python tools/knn_probe.py 768 f16 20 1000000 16 2>&1 | grep -v amdgpu.ids | sed "s/^/emit_kl32=0: /" | tee -a $O/knn_kl32.log
TOOLS_DEV=1 VIDMEM_EMIT_KL32=1 python tools/knn_probe.py 768 f16 20 1000000 16 2>&1 | grep -v amdgpu.ids | sed "s/^/emit_kl32=1: /" | tee -a $O/knn_kl32.log
python -m pytest tests/test_topk_gpu.py -q > $O/topk_tests.log 2>&1; trc=$?
tail -4 $O/topk_tests.log
python tools/uncert_probe.py 16 2>&1 | grep -v amdgpu.ids | tee $O/uncert16.log | head -30
for a in 0 1 2 4 8 16 32 3 10 48 63 0; do
  TOOLS_DEV=1 VIDMEM_ATTN_ABL=$a python tools/enc_probe.py clip_l14_336 bf16 224 2 3 2>&1 | grep "attention" | sed "s/.*'attention': (\([0-9.]*\), \([0-9]*\)).*/abl $a: attention \1 ms in \2 launches/" | tee -a $O/attn_abl.log
done
exit $trc
