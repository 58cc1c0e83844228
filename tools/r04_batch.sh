export CDB_BENCH_TRACE=1
fails=0
for i in $(seq 1 40); do
python bench.py --gpus 2 --backend gloo --share-gpu --workload mid --steps 2 --warmup 1 --configs none --no-cpu-baseline > gpurun_out/r04_sl_x.json 2> gpurun_out/r04_sl_x.err; rc=$?; fb=$(grep -c 'group_fallbacks 1' gpurun_out/r04_sl_x.err); if [ $rc != 0 ] || [ $fb != 0 ]; then echo "run $i rc=$rc fallbacks=$fb"; grep "fault\|rror" gpurun_out/r04_sl_x.err | tail -4; fi; if [ $rc != 0 ]; then fails=$((fails+1)); fi
done
echo "done t=$SECONDS fails=$fails"
