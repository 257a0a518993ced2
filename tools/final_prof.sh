set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2v; mkdir -p $O
python $R/bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline > $O/stats.log 2>&1; echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-streaming --no-c3 --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-streaming --no-c3 --steps 3 --warmup 1 > $O/pmc_write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --no-cpu-baseline --no-streaming --no-c3 --steps 3 --warmup 1 > $O/pmc_mfma.log 2>&1; echo "mfma rc=$?"
ls -R $O | head -40
