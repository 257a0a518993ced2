#!/bin/bash
# round 4, GPU call 15: whole GPU suite, default bench line, the round's profiles (tools/r4_prof.sh), and the 2-rank gloo
# rehearsal of the N > 1 path on the one GPU
set -o pipefail
O=gpurun_out/r4z
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; trc=$?; echo "tests rc=$trc"; tail -3 $O/gpu_tests.log
[ $trc -ne 0 ] && exit $trc
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python tools/benchsum2.py $O/bench.json 2>&1 | head -60
bash tools/r4_prof.sh r4z > $O/prof.log 2>&1; echo "prof rc=$?"; tail -3 $O/prof.log
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"; tail -c 600 $O/bench_gloo2.json
