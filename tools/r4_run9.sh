#!/bin/bash
# round 4, GPU call 9: the CU store path on the GEMM epilogue's pattern (tools/store_probe.hip): all CUs of an XCD at once
# against every 2nd / 4th / 8th / 32nd, nt against plain, three row strides
O=gpurun_out/r4i; mkdir -p $O
B=tools/bin
for e in 1 2 4 8 32; do for nt in 1 0; do timeout -k 5 60 $B/store_probe $e $nt 16 2304 | tee -a $O/store.log; done; done
for ldo in 3072 768; do for e in 1 8; do timeout -k 5 60 $B/store_probe $e 1 16 $ldo | tee -a $O/store.log; done; done
exit 0
