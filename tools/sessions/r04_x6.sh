#!/bin/bash
O=gpurun_out/r04_x6; mkdir -p $O
bash tools/prof.sh r04_x6/prof > $O/prof_head.txt 2>&1
db=$(find $O/prof -name "*.db" | head -1); python tools/forward_timeline.py $db > $O/timeline.txt 2>&1; head -n 1 $O/timeline.txt; tail -n 1 $O/timeline.txt
rm -rf $O/prof/prof
