#!/bin/bash
# round 4, GPU call 2: table GELU vs the polynomial GELU in the GEMM harness (alternating), kernels bit for bit against
# each other, encoder parity tests, and the schedule A/B through bench.py's main leg
set -o pipefail
O=gpurun_out/r4b
mkdir -p $O
B=tools/bin
# FC1 of one 880-frame pass and of a 441-frame pass: table / poly alternating, three rounds
for r in 1 2 3; do
  $B/gemm_bench 173360 3072 768 1 20 2>&1 | grep -v amdgpu.ids | sed "s/^/table r$r: /" >> $O/gelu_ab.log
  $B/gemm_bench_poly 173360 3072 768 1 20 2>&1 | grep -v amdgpu.ids | sed "s/^/poly  r$r: /" >> $O/gelu_ab.log
done
cat $O/gelu_ab.log
# the three kernels against each other (same table code in all): bits must not differ
VARIANTS=3,1,2 $B/gemm_bench 5000 3072 768 1 5 > $O/gelu_bits.log 2>&1; cat $O/gelu_bits.log | grep -v amdgpu.ids
VARIANTS=3,1,2 $B/gemm_bench 86877 2304 768 0 10 > $O/store_bits.log 2>&1; cat $O/store_bits.log | grep -v amdgpu.ids
ZERO=1 $B/gemm_bench 173360 3072 768 1 20 2>&1 | grep -v amdgpu.ids | sed "s/^/table ZERO: /"
ZERO=1 $B/gemm_bench_poly 173360 3072 768 1 20 2>&1 | grep -v amdgpu.ids | sed "s/^/poly  ZERO: /"
python -m pytest tests/test_encoder_gpu.py tests/test_dropin_gpu.py -q > $O/enc_tests.log 2>&1; trc=$?
tail -5 $O/enc_tests.log; grep -h "rel err\|vs quant" $O/enc_tests.log | head -20
L="--no-knn --no-c4 --no-extractor --no-streaming --no-c3 --no-cpu-baseline --no-ceiling --no-rccl-world1 --no-profile"
for r in 1 2; do
  python bench.py $L --schedule one_stream > $O/b_one_$r.json 2>/dev/null && python tools/show_bench.py "one_stream 1760 r$r" < $O/b_one_$r.json | cut -c1-60
  python bench.py $L > $O/b_auto_$r.json 2>/dev/null && python tools/show_bench.py "auto 1760 r$r" < $O/b_auto_$r.json | cut -c1-60
  python bench.py $L --chunks-per-step 55 > $O/b_880_$r.json 2>/dev/null && python tools/show_bench.py "one pass 880 r$r" < $O/b_880_$r.json | cut -c1-60
done
exit $trc
