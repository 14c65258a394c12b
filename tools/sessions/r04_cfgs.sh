#!/bin/bash
# round 4, final build: configs[3] / configs[4] benches and the 16-prompt bf16 reference of the fp8 line
O=gpurun_out/r04_cfgs; mkdir -p $O
python bench.py --config c4 --steps 4 --warmup 2 --no-pmc > $O/bench_c4.log 2>&1; tail -n 1 $O/bench_c4.log | cut -c1-200
python bench.py --config c5 --steps 4 --warmup 2 --no-pmc > $O/bench_c5.log 2>&1; tail -n 1 $O/bench_c5.log | cut -c1-200
python bench.py --prompts-per-gpu 16 --steps 4 --warmup 2 --no-pmc --no-cpu-baseline > $O/bench_c2_16.log 2>&1; tail -n 1 $O/bench_c2_16.log | cut -c1-200
