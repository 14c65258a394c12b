#!/bin/bash
# r05 session 22: final build (A/B switches of the round removed): GPU suite, smoke, driver bench command
O=gpurun_out/r05_s22; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; tail -2 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1; tail -1 $O/bench_driver_cmd.log | cut -c1-260
