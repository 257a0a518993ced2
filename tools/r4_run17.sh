#!/bin/bash
O=gpurun_out/r4r; mkdir -p $O
for l in 95c37f5 114768f; do TOOLS_LIB=tools/bin/libvidmem_$l.so python tools/golden_probe.py 2>&1 | grep -v amdgpu.ids | grep "rep 0" | sed "s/^/$l: /" | tee -a $O/golden.log; done
python tools/golden_probe.py 2>&1 | grep -v amdgpu.ids | grep "rep 0" | sed "s/^/HEAD: /" | tee -a $O/golden.log
exit 0
