# Round-4 profiles (run through gpurun): rocprofv3 kernel-trace summary of the main leg on the ONE-STREAM schedule (the
# schedule roofline.avg_launch_ms is taken on: with two streams a kernel's duration includes waiting for the other
# stream), then the counter passes - FETCH_SIZE / WRITE_SIZE / matrix-pipe utilisation - one counter set per run, the
# program directly after `--` (MI355X_MICROARCH.md).  Summaries land in gpurun_out/<tag>/ and are copied to profiles/.
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-r4p}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
LEGS="--no-cpu-baseline --no-streaming --no-extractor --no-knn --no-c4 --no-c3 --no-ceiling --no-rccl-world1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $LEGS --schedule one_stream > $O/bench_main_leg_one_stream.json 2> $O/stats.log; echo "stats rc=$?"
S=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$S" $O/bench_main_leg_kernel_stats.csv; rm -rf $O/stats
A="--no-cpu-baseline --no-streaming --no-extractor --no-ceiling --no-rccl-world1 --steps 3 --warmup 1 --c3-frames 448 --no-two-stream"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $A > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py $A > $O/pmc_write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py $A > $O/pmc_mfma.log 2>&1; echo "mfma rc=$?"
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1); M=$(find $O/pmc_mfma -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_traffic.py $F $W $O/pmc_traffic.json main=F1760,mb883,R100000,k10 c3=mb224 > $O/pmc_traffic.txt; echo "fold traffic rc=$?"
python3 $R/tools/pmc_mfma.py $M $O/pmc_mfma.json "bench.py $A" > $O/pmc_mfma.txt; echo "fold mfma rc=$?"
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma
head -30 $O/bench_main_leg_kernel_stats.csv; cat $O/pmc_traffic.txt $O/pmc_mfma.txt | head -60
