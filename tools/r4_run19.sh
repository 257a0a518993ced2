#!/bin/bash
O=gpurun_out/r4t; mkdir -p $O
python bench.py --no-knn --no-c4 --no-c3 --no-streaming --no-cpu-baseline --no-ceiling --no-rccl-world1 > $O/bench_ext.json 2> $O/bench_ext.err; echo "rc=$?"; tail -2 $O/bench_ext.err; python tools/benchsum2.py $O/bench_ext.json | grep -E "value|extractor"
