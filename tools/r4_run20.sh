#!/bin/bash
O=gpurun_out/r4u; mkdir -p $O
python tools/extractor_profile.py 16384 0 1 > $O/prof16k.log 2>&1; grep -v amdgpu.ids $O/prof16k.log | head -60
