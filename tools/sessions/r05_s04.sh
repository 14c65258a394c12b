#!/bin/bash
# r05 session 4 (re-entry baseline): driver bench command, rocprofv3 kernel stats, forward timeline on the head build
O=gpurun_out/r05_s04; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1; tail -1 $O/bench_driver_cmd.log | cut -c1-600
bash tools/prof.sh r05_s04/prof > $O/prof_head.txt 2>&1
db=$(find $O/prof -name "*.db" | head -1); python tools/forward_timeline.py $db > $O/forward_timeline.txt 2>&1; head -3 $O/forward_timeline.txt
cp $O/prof/kernel_stats.md $O/kernel_stats.md
rm -rf $O/prof/prof
