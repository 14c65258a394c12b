mkdir -p gpurun_out/r02h
timeout 2200 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r02h/gpu_tests.log; tail -2 gpurun_out/r02h/gpu_tests.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02h/bench_final3.log 2>&1; tail -1 gpurun_out/r02h/bench_final3.log | cut -c1-400
bash tools/prof.sh r02h > gpurun_out/r02h/prof_head.txt 2>&1; head -24 gpurun_out/r02h/prof_head.txt
