#!/bin/bash
# Developer helper (GPU box): A/B two builds of libvidmem.so on bench.py, alternating, same box.
#   tools/ab_bench.sh <pkg>/libvidmem_prev.so [rounds]
P=real-time-brain-inspired-video-memory_amd
OTHER=$1; N=${2:-2}
mkdir -p gpurun_out/ab
cp $P/libvidmem.so /tmp/new.so
for i in $(seq $N); do
  for w in new prev; do
    if [ $w = new ]; then cp /tmp/new.so $P/libvidmem.so; else cp $OTHER $P/libvidmem.so; fi
    python bench.py --no-cpu-baseline --no-streaming --no-c3 --steps 6 --warmup 2 > gpurun_out/ab/$w$i.json 2> gpurun_out/ab/err.log || exit 1
    echo "== $w"; python tools/benchsum.py gpurun_out/ab/$w$i.json
  done
done
cp /tmp/new.so $P/libvidmem.so
