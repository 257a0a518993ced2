#!/bin/bash
# round 4, GPU call 26: FC1's instantiation at 224 VGPRs (bias ring of four, table two blocks ahead, one-scalar-constant
# address arithmetic): bit identity across the three GEMM kernels, harness time against the poly build, encoder parity,
# then alternating A/B of three library builds (prev = 227/244, head = 224/244, head2 = 224/224)
set -o pipefail
O=gpurun_out/r4B; mkdir -p $O
B=tools/bin
VARIANTS=3,1,2 timeout -k 5 120 $B/gemm_bench 5000 3072 768 1 5 2>&1 | grep -v amdgpu.ids | tee $O/bits.log
VARIANTS=3,1,2 timeout -k 5 120 $B/gemm_bench 70000 3072 768 1 5 2>&1 | grep -v amdgpu.ids | tee -a $O/bits.log
for r in 1 2; do timeout -k 5 120 $B/gemm_bench 173360 3072 768 1 20 2>&1 | grep variant | sed "s/maxerr.*//" | tee -a $O/fc1.log; done
python -m pytest tests/test_encoder_gpu.py -q -x > $O/tests.log 2>&1; trc=$?
tail -3 $O/tests.log
[ $trc -ne 0 ] && exit $trc
for r in 1 2 3; do for l in prev head head2; do
  TOOLS_LIB=tools/bin/libvidmem_$l.so python tools/enc_probe.py vit_b16_224 f16 1760 6 2>&1 | grep "frames/s" | sed "s/^/$l: /" | tee -a $O/ab.log
done; done
exit 0
