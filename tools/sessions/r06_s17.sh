set +x
# final build: full GPU suite + the driver's command + soak
O=gpurun_out/r06_s17; mkdir -p $O
python -m pytest tests -m gpu -q -s > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1; tail -1 $O/bench_driver_cmd.log | cut -c1-300
python tools/soak.py > $O/soak.log 2>&1; tail -3 $O/soak.log
