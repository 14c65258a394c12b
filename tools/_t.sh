for t in 24 16 12 20 24 16; do
GILL_GEMM_MINSTEPS=$t timeout 600 python bench.py --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('minsteps $t', r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'])"
done
