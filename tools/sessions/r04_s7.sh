#!/bin/bash
# round 4, GPU session 7: split-K reducer + fused GroupNorm — operator tests, engine switch test, oracle tests, loop A/B, launch count
O=gpurun_out/r04_s7; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -k "fused_groupnorm or conv3x3" -q -s > $O/ops.log 2>&1; tail -n 3 $O/ops.log; grep "conv+GN" $O/ops.log
python -m pytest tests/test_configs_gpu.py -k "switches and RED_GN" -q -s > $O/switch.log 2>&1; tail -n 3 $O/switch.log
python -m pytest tests/test_configs_gpu.py -k "teacher" -q -s > $O/teacher.log 2>&1; tail -n 3 $O/teacher.log
python -m pytest tests/test_soak_gpu.py -q > $O/soak.log 2>&1; tail -n 1 $O/soak.log
bash tools/ab_env.sh GILL_GEMM_RED_GN 3 > $O/ab.log 2>&1; cat $O/ab.log
bash tools/prof.sh r04_s7/prof > $O/prof_head.txt 2>&1
db=$(find $O/prof -name "*.db" | head -1); python tools/forward_timeline.py $db > $O/timeline.txt 2>&1; head -n 1 $O/timeline.txt; tail -n 1 $O/timeline.txt
rm -rf $O/prof/prof
