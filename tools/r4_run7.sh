#!/bin/bash
# round 4, GPU call 7 (after the container was re-created): whole GPU suite, the default bench line, then the round's profiles
set -o pipefail
O=gpurun_out/r4g
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/gpu_tests.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python tools/benchsum2.py $O/bench.json 2>&1 | head -60
bash tools/r4_prof.sh r4g > $O/prof.log 2>&1; echo "prof rc=$?"; tail -5 $O/prof.log
