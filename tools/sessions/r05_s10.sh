#!/bin/bash
# r05 session 10: full GPU suite + driver bench command on the build with STREAM64 (8 waves) and the switch / PP_ABL clean-up
O=gpurun_out/r05_s10; mkdir -p $O
python -m pytest tests -m gpu -q -s > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1; tail -1 $O/bench_driver_cmd.log | cut -c1-300
