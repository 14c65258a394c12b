#!/bin/bash
# round 4: the whole GPU suite (timed) + C1 end to end on the box's host cores
O=gpurun_out/r04_tests; mkdir -p $O
S=$(date +%s); python -m pytest tests -m gpu -q -s > $O/gpu_tests.log 2>&1; E=$(date +%s); tail -n 3 $O/gpu_tests.log; echo "GPU suite wall time: $((E-S)) s" | tee $O/gpu_tests_time.log
grep -E "FAILED|ERROR" $O/gpu_tests.log | head
python -m pytest tests -m gpu -q --durations=15 -x > $O/gpu_tests_durations.log 2>&1; grep -A 18 "slowest" $O/gpu_tests_durations.log | head -20
python tools/c1_cpu_end_to_end.py --steps 10 > $O/c1_cpu.log 2>&1; tail -n 1 $O/c1_cpu.log
