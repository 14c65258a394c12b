timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x 2>&1 | tail -1
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('new', r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'])"
GILL_AMD_LIB=/root/repo/tools/_lib_prev.so timeout 600 python bench.py --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('prev', r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'])"
done
