#!/bin/bash
O=gpurun_out/r4C; mkdir -p $O
python bench.py --no-knn --no-c4 --no-c3 --no-extractor --no-cpu-baseline --no-ceiling --no-rccl-world1 > $O/bench_stream.json 2> $O/bench.err; echo "rc=$?"; python tools/benchsum2.py $O/bench_stream.json | grep -E "^value|streaming|feed"
