#!/bin/bash
O=gpurun_out/r4F; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
