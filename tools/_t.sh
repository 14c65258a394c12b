timeout 900 python -m pytest tests/test_fp8_gpu.py -q -m gpu -s -k unet_mode 2>&1 | grep -E "fp8 UNet|passed|failed|Error|assert|rror" | head -30
timeout 600 python bench.py --config c5 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['output_check'])"
timeout 600 python bench.py --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'])"
