#!/bin/bash
# round 4, last GPU call: the default bench line with the final bench.py (the profiles of tools/r4_run15.sh were taken on
# the same library; bench.py gained the feed breakdown and the long extractor clip since)
O=gpurun_out/r4D; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python tools/benchsum2.py $O/bench.json 2>&1 | head -40
python -m pytest tests -m gpu -x -q -k "smoke or abi or topk or dropin" 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
