set +x
O=gpurun_out/r06_s23; mkdir -p $O
python -m pytest tests -m gpu -q -s > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1; tail -1 $O/bench_driver_cmd.log | cut -c1-300
python bench.py --config c4 --steps 4 --warmup 2 --no-pmc > $O/bench_c4.log 2>&1; tail -n 1 $O/bench_c4.log | cut -c1-200
python bench.py --prompts-per-gpu 16 --steps 4 --warmup 2 --no-pmc --no-cpu-baseline > $O/bench_c2_16.log 2>&1; tail -n 1 $O/bench_c2_16.log | cut -c1-200
