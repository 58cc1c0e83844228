cd tools/experiments
for b in seg_bench_nr seg_bench_i15_w8_true seg_bench_i15_w8_false seg_bench_i14_w8_true seg_bench_i12_w8_true seg_bench_i15_w1_true; do echo "== $b"; timeout 120 ./$b 30 256 3 2>&1 | sed -n 2p; done
