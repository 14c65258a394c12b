timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_stages_gpu.py -q -m gpu -x 2>&1 | tail -2
for t in 1 0 1 0; do
GILL_GEMM_STAGED_OFF=$t timeout 600 python bench.py --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('staged_off $t', r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'])"
done
export GILL_OP_REPEAT=100
for K in 64 320 1280; do python tools/one_op.py gemm 32768 320 $K 1 2>&1 | tail -1; GILL_GEMM_STAGED_OFF=1 python tools/one_op.py gemm 32768 320 $K 1 2>&1 | tail -1; done
