python -m pytest tests -m gpu -q -x > /tmp/t.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR|Error|assert " /tmp/t.log | head -30 > gpurun_out/r04_gputests4.log; tail -c 1500 gpurun_out/r04_gputests4.log
bash tools/profile_round.sh r04d d025989 2>&1 | grep -E "^===|rs_sweep|textgen|GiB" | head -40
