# Cache-policy probe of the persistent GEMM's operand loads (run through gpurun): harness binaries built with
# -DVM_GEMM_X_AUX / -DVM_GEMM_W_AUX (tools/bin/gemm_bench_{a: default, b: X nt, c: W nt, d: both}), FC1 at the bench
# micro-batch (173,360 rows), time without the profiler and FETCH_SIZE / WRITE_SIZE with it.
# Build (in the container, from the repo root; P = the package directory):
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/gemm_bench.hip -o /tmp/gemm_bench.o
#   for v in "0 0 a" "2 0 b" "0 2 c" "2 2 d"; do set -- $v
#     hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DVM_GEMM_ABLATE -DVM_GEMM_X_AUX=$1 \
#           -DVM_GEMM_W_AUX=$2 -Iinclude -c $P/csrc/gemm.hip -o /tmp/gemm_abl_$3.o
#     hipcc --offload-arch=gfx950 /tmp/gemm_bench.o /tmp/gemm_abl_$3.o $P/csrc/context.o -o tools/bin/gemm_bench_$3; done
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pol}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SH="173360 3072 768 1 20"
for v in a b c d a b; do echo "== $v"; timeout -k 10 120 $R/tools/bin/gemm_bench_$v $SH | tee -a $O/time_$v.txt || exit 1; done
for v in a b c d; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/p_${v}_$c -- $R/tools/bin/gemm_bench_$v 173360 3072 768 1 4 > $O/p_${v}_$c.log 2>&1 || exit 1
    F=$(find $O/p_${v}_$c -name "*counter_collection.csv" | head -1)
    python3 - $F $v $c <<'EOF'
import csv, statistics, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "gemm256p" in r["Kernel_Name"]]
m = statistics.median(v) * 1024 * (2 if sys.argv[3] == "FETCH_SIZE" else 1)   # KiB units; reads doubled on gfx950 (guide)
print(f"{sys.argv[2]} {sys.argv[3]} launches {len(v)}: {m / 1e6:.0f} MB per launch")
EOF
    rm -rf $O/p_${v}_$c
  done
done
