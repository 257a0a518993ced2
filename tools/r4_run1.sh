#!/bin/bash
# round 4, GPU call 1: the whole GPU suite, the schedule A/B at the two candidate step sizes, one bench run
set -o pipefail
O=gpurun_out/r4a
mkdir -p $O
python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; trc=$?
tail -8 $O/gpu_tests.log
python tools/dual_probe.py 1760 vit_b16_224 f16 one_stream,two_streams 0 > $O/dual_1760.log 2>&1 && tail -6 $O/dual_1760.log &&
python tools/dual_probe.py 880 vit_b16_224 f16 one_stream,two_streams 440 > $O/dual_880_mb440.log 2>&1 && tail -6 $O/dual_880_mb440.log &&
python tools/dual_probe.py 880 vit_b16_224 f16 one_stream 0 > $O/single_880.log 2>&1 && tail -3 $O/single_880.log &&
python bench.py > $O/bench.json 2> $O/bench.err; rc=$?
tail -3 $O/bench.err
python tools/benchsum2.py $O/bench.json
[ $trc -ne 0 ] && exit $trc
exit $rc
