timeout 1500 python -m pytest tests/test_stages_gpu.py tests/test_coverage_gpu.py -q -m gpu -x -k "unet or sd or denoise or vae or F8 or full" 2>&1 | tail -3
for t in 1 0 1 0; do
if [ $t = 1 ]; then export GILL_CONV_OUT_DIRECT=1; else unset GILL_CONV_OUT_DIRECT; fi
timeout 600 python bench.py --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('direct $t', r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac'])"
done
