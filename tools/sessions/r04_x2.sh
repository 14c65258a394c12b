#!/bin/bash
O=gpurun_out/r04_x2; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "cross_attention_folded" -s > $O/ops.log 2>&1; echo "ops rc=$?"; tail -n 2 $O/ops.log
bash tools/prof.sh r04_x2/prof > $O/prof_head.txt 2>&1
db=$(find $O/prof -name "*.db" | head -1); python tools/forward_timeline.py $db > $O/timeline.txt 2>&1; head -n 1 $O/timeline.txt; tail -n 1 $O/timeline.txt
rm -rf $O/prof/prof
