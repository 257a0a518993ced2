#!/bin/bash
# round 4, GPU call 8: where the persistent GEMM's time goes on the four encoder shapes of one 880-frame pass (173,360 rows):
# whole kernel / no stores / no epilogue, wall time and per-dispatch cycles + matrix-pipe utilisation (PMC); feature-group
# budgets of the tile order (groups that do not straddle an XCD's 32 workgroups)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4h; mkdir -p $O
B=$R/tools/bin
cd /tmp && export TMPDIR=/tmp
shape() { # name N K epi
  for r in 1 2; do VARIANTS=3,1032,1040 timeout -k 10 120 $B/gemm_bench 173360 $2 $3 $4 20 2>&1 | grep "variant" | sed "s/^/$1 r$r: /; s/maxerr.*//" | tee -a $O/abl_time.log; done
  VARIANTS=3,1032,1040 timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p_$1 -- $B/gemm_bench 173360 $2 $3 $4 8 > $O/p_$1.log 2>&1 || { echo "pmc $1 failed"; tail -5 $O/p_$1.log; return 1; }
  F=$(find $O/p_$1 -name "*counter_collection.csv" | head -1)
  python3 $R/tools/pmc_harness.py $F | sed "s/^/$1: /" | tee -a $O/abl_cycles.log
  rm -rf $O/p_$1
}
shape qkv 2304 768 0 && shape fc1 3072 768 1 && shape proj 768 768 0 && shape fc2 768 3072 0
for kb in 2560 1600 3200 2560 1600 3200; do
  VIDMEM_GEMM_WGROUP_KB=$kb timeout -k 10 120 $B/gemm_bench 173360 3072 768 1 20 2>&1 | grep "variant" | sed "s/^/fc1 wgroup_kb $kb: /; s/maxerr.*//" | tee -a $O/fgroup.log
done
for kb in 2560 1300 2560 1300; do
  VIDMEM_GEMM_WGROUP_KB=$kb timeout -k 10 120 $B/gemm_bench 173360 2304 768 0 20 2>&1 | grep "variant" | sed "s/^/qkv wgroup_kb $kb: /; s/maxerr.*//" | tee -a $O/fgroup.log
done
exit 0
