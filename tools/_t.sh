timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_stages_gpu.py -q -m gpu -x 2>&1 | tail -2
for t in 0 1 0 1; do
GILL_GEMM_PP=$t timeout 600 python bench.py --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('pp $t', r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['output_check']['all_finite'])"
done
