for t in 0 12 24 48 0 12; do
GILL_GEMM_WTOUCH=$t timeout 600 python bench.py --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('wtouch $t', r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'])"
done
GILL_GEMM_WTOUCH=12 timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -x 2>&1 | tail -2
