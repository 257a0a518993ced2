#!/bin/bash
# round 4, GPU call 11: the persistent GEMM's 16-bit epilogue without the LDS transpose (lane-row swaps) against the LDS
# version: store patterns alone (probe), bit-identity across kernels, alternating A/B with tile-boundary stamps
O=gpurun_out/r4l; mkdir -p $O
B=tools/bin
for pat in 0 2; do for e in 1 8 32; do timeout -k 5 60 $B/store_probe $e 1 16 2304 -1 $pat | tee -a $O/store_pat.log; done; done
for sw in 1 0; do
  VIDMEM_GEMM_EPI_SWAP=$sw VARIANTS=3,1,2 timeout -k 5 120 $B/gemm_bench 5000 2304 768 0 5 2>&1 | grep -v amdgpu.ids | sed "s/^/swap $sw: /" | tee -a $O/bits.log
  VIDMEM_GEMM_EPI_SWAP=$sw VARIANTS=3,1,2 timeout -k 5 120 $B/gemm_bench 5000 3072 768 1 5 2>&1 | grep -v amdgpu.ids | sed "s/^/swap $sw: /" | tee -a $O/bits.log
  VIDMEM_GEMM_EPI_SWAP=$sw VARIANTS=3,2 timeout -k 5 120 $B/gemm_bench 1000 768 3072 5 5 2>&1 | grep -v amdgpu.ids | sed "s/^/swap $sw: /" | tee -a $O/bits.log
done
ab() { for r in 1 2; do for sw in 0 1; do STAMPS=1 VIDMEM_GEMM_EPI_SWAP=$sw timeout -k 5 120 $B/gemm_bench 173360 $1 $2 $3 20 2>&1 | grep -E "variant|round  8|round 20" | sed "s/maxerr.*//; s/K-loop-done.*epilogue/epilogue/" | sed "s/^/swap $sw: /" | tee -a $O/ab.log; done; done; }
ab 2304 768 0
ab 3072 768 1
ab 768 768 0
ab 768 3072 0
exit 0
