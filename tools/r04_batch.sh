python -m pytest tests -m gpu -q -x > /tmp/t.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR|Error|assert " /tmp/t.log | head -30 > gpurun_out/r04_gputests6.log; tail -c 1500 gpurun_out/r04_gputests6.log
bash tools/profile_round.sh r04d 622374e 2>&1 | grep -E "^===|rs_sweep|sa_round|GiB" | head -40
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
