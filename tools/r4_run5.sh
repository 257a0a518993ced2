#!/bin/bash
# round 4, GPU call 5: the whole GPU suite on the current tree, the 577-token attention after the chunked fill,
# the top-20 bf16 search after the cascade change, one full bench run
set -o pipefail
O=gpurun_out/r4e
mkdir -p $O
python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; trc=$?
tail -8 $O/gpu_tests.log
for i in 1 2; do python tools/enc_probe.py clip_l14_336 bf16 224 2 3 2>&1 | grep "attention" | sed "s/.*'attention': (\([0-9.]*\), \([0-9]*\)).*/attention \1 ms in \2 launches/" | tee -a $O/attn.log; done
python tools/knn_probe.py 1024 bf16 20 1000000 16,32 2>&1 | grep -v amdgpu.ids | tee $O/knn.log
python tools/knn_probe.py 768 f16 10 1000000 16,64,256 2>&1 | grep -v amdgpu.ids | tee -a $O/knn.log
python bench.py > $O/bench.json 2> $O/bench.err; rc=$?
tail -3 $O/bench.err
python tools/benchsum2.py $O/bench.json
[ $trc -ne 0 ] && exit $trc
exit $rc
