#!/bin/bash
# round 4, GPU call 24: GEMM back at 224 VGPRs (the low-register LayerNorm of the other stream fits beside it again):
# encoder parity tests, then the main leg twice (two streams against the one-stream repeat) and the C3 leg
set -o pipefail
O=gpurun_out/r4y; mkdir -p $O
python -m pytest tests/test_encoder_gpu.py -q -x > $O/tests.log 2>&1; trc=$?
tail -3 $O/tests.log
[ $trc -ne 0 ] && exit $trc
L="--no-knn --no-c4 --no-extractor --no-streaming --no-cpu-baseline --no-rccl-world1"
for r in 1 2; do python bench.py $L > $O/bench$r.json 2> $O/bench.err; python tools/benchsum2.py $O/bench$r.json | grep -E "^value|one-stream|roofline|mfma_ceiling|^c3:"; done
exit 0
