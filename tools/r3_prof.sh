# Round-3 counter passes (run through gpurun): FETCH_SIZE / WRITE_SIZE / matrix-pipe utilisation of the main leg, the C4
# rank-share leg and the C3 leg.  One counter set per rocprofv3 run, program directly after `--` (MI355X_MICROARCH.md).
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-r3p}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--no-cpu-baseline --no-streaming --no-extractor --steps 3 --warmup 1 --c3-frames 448 --no-two-stream"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $A > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py $A > $O/pmc_write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py $A > $O/pmc_mfma.log 2>&1; echo "mfma rc=$?"
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1); M=$(find $O/pmc_mfma -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_traffic.py $F $W $O/pmc_traffic.json main=F880,mb880,R100000,k10 c3=mb224 > $O/pmc_traffic.txt; echo "fold traffic rc=$?"
python3 $R/tools/pmc_mfma.py $M $O/pmc_mfma.json "bench.py $A" > $O/pmc_mfma.txt; echo "fold mfma rc=$?"
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma
cat $O/pmc_traffic.txt $O/pmc_mfma.txt
