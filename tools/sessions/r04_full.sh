#!/bin/bash
# round 4: the whole GPU suite (timed), the driver's bench command, configs[3] / configs[4] benches
O=gpurun_out/r04_full; mkdir -p $O
/usr/bin/time -v python -m pytest tests -m gpu -q -s > $O/gpu_tests.log 2> $O/gpu_tests_time.log; tail -n 3 $O/gpu_tests.log; grep "Elapsed" $O/gpu_tests_time.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1; tail -n 1 $O/bench_driver_cmd.log | cut -c1-600
python bench.py --config c4 --steps 4 --warmup 2 --no-pmc > $O/bench_c4.log 2>&1; tail -n 1 $O/bench_c4.log | cut -c1-300
python bench.py --config c5 --steps 4 --warmup 2 --no-pmc > $O/bench_c5.log 2>&1; tail -n 1 $O/bench_c5.log | cut -c1-300
python bench.py --prompts-per-gpu 16 --steps 4 --warmup 2 --no-pmc --no-cpu-baseline > $O/bench_c2_16.log 2>&1; tail -n 1 $O/bench_c2_16.log | cut -c1-300
