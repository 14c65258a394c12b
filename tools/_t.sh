mkdir -p gpurun_out/r02
timeout 900 python bench.py --config c5 --no-cpu-baseline --no-pmc > gpurun_out/r02/bench_c5.log 2>&1; tail -1 gpurun_out/r02/bench_c5.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('c5 P16', r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['output_check'])"
timeout 900 python bench.py --prompts-per-gpu 16 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('bf16 P16', r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'])"
