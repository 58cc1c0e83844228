cd tools/experiments
for b in seg_bench_nr seg_bench_nt512_lb4 seg_bench_nt512_lb8 seg_bench_nt256_lb4; do echo "== $b"; timeout 120 ./$b 30 256 3 2>&1 | tail -4; done
