#!/bin/bash
O=gpurun_out/r4q; mkdir -p $O
python tools/golden_probe.py 2>&1 | grep -v amdgpu.ids | sed "s/^/release: /" | tee -a $O/golden.log
TOOLS_DEV=1 VIDMEM_ATTN_CTX_NT=0 python tools/golden_probe.py 2>&1 | grep -v amdgpu.ids | sed "s/^/dev ctx_nt=0: /" | tee -a $O/golden.log
TOOLS_DEV=1 VIDMEM_ATTN_CTX_NT=1 python tools/golden_probe.py 2>&1 | grep -v amdgpu.ids | sed "s/^/dev ctx_nt=1: /" | tee -a $O/golden.log
python -m pytest tests/test_encoder_gpu.py -q -k "full_models" 2>&1 | tail -3
exit 0
