#!/bin/bash
# poll clocks / power / temperature while the bench runs
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power|Temperature" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.5; done ) > gpurun_out/smi_poll.txt &
python bench.py --no-cpu-baseline --no-other-workloads --min-seconds 8 --max-repeats 4000 > gpurun_out/bench_long.json 2>/dev/null
wait
