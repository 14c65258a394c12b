timeout 1500 python -m pytest tests/test_stages_gpu.py tests/test_coverage_gpu.py tests/test_fp8_gpu.py -q -m gpu -x -k "unet or sd or denoise or fp8 or F8 or c1 or full" 2>&1 | tail -3
for t in 1 0 1 0; do
if [ $t = 1 ]; then export GILL_UNET_FFO_UNFUSED=1; else unset GILL_UNET_FFO_UNFUSED; fi
timeout 600 python bench.py --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('unfused $t', r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'])"
done
