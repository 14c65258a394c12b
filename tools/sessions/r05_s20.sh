#!/bin/bash
# r05 session 20: clocks and power while the denoise loop runs (rocm-smi sampled every 0.5 s next to a bench run)
O=gpurun_out/r05_s20; mkdir -p $O
rocm-smi --showclocks --showpower --showmaxpower > $O/idle.txt 2>&1
(python bench.py --steps 40 --warmup 2 --no-pmc --no-scale-origin --no-cpu-baseline > $O/bench.log 2>&1) &
BP=$!
sleep 9
for i in $(seq 1 16); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.5; done > $O/load.txt
wait $BP
tail -4 $O/idle.txt; sed "s/=*;//g" $O/load.txt | cut -c1-400 | head -16
