#!/bin/bash
O=gpurun_out/r4w; mkdir -p $O
python -m pytest tests/test_dropin_gpu.py tests/test_ingest_gpu.py -q -x 2>&1 | tail -3
for r in 1 2; do python bench.py --no-knn --no-c4 --no-c3 --no-streaming --no-cpu-baseline --no-ceiling --no-rccl-world1 > $O/bench_ext$r.json 2> $O/bench_ext.err; python tools/benchsum2.py $O/bench_ext$r.json | grep -E "value|extractor"; done
