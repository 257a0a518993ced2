#!/bin/bash
# round 4, GPU call 10: (a) the store path of ONE XCD alone (per-XCD or chip-wide limit?); (b) tile-boundary stamps of the
# persistent GEMM: how far apart do the 256 CUs run, with and without a start ramp, and how long is an epilogue?
O=gpurun_out/r4j; mkdir -p $O
B=tools/bin
for e in 1 2 4; do timeout -k 5 60 $B/store_probe $e 1 16 2304 0 | tee -a $O/store_xcd.log; done
timeout -k 5 60 $B/store_probe 1 0 16 2304 0 | tee -a $O/store_xcd.log
st() { STAMPS=1 VIDMEM_GEMM_DEPHASE_RAMP=$1 VIDMEM_GEMM_DEPHASE_PCT=$2 timeout -k 5 120 $B/gemm_bench 173360 $3 $4 $5 10 2>&1 | grep -v amdgpu.ids | sed "s/maxerr.*//" | sed "s/^/ramp $1 pct $2: /" | tee -a $O/stamps.log; }
st 0 0 2304 768 0
st 40 0 2304 768 0
st 0 100 2304 768 0
st 0 0 3072 768 1
st 80 0 3072 768 1
st 0 0 768 3072 0
exit 0
