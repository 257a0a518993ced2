# Round-3 evidence set (run through gpurun): the bench line, rocprofv3 per-kernel stats of the same command, PMC passes.
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-r3v}; O=$R/gpurun_out/$T; mkdir -p $O
python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-two-stream --stream-replays 220 > $O/stats.log 2>&1; echo "stats rc=$?"
S=$(find /tmp/stats -name "*kernel_stats.csv" | head -1); cp "$S" $O/bench_kernel_stats.csv; rm -rf /tmp/stats
head -12 $O/bench_kernel_stats.csv
bash $R/tools/r3_prof.sh $T
