timeout 900 python -m pytest tests/test_fp8_gpu.py -q -m gpu -x 2>&1 | tail -2
export GILL_OP_REPEAT=50
for sh in "8 64 64 320 0 320 1" "8 64 64 640 0 320 1" "8 32 32 640 0 640 2" "8 16 16 1280 0 1280 4"; do
  GILL_GEMM_PP=0 python tools/one_op.py conv8 $sh 2>&1 | tail -1
  GILL_GEMM_PP=1 python tools/one_op.py conv8 $sh 2>&1 | tail -1
done
for t in 0 1 0 1; do
GILL_GEMM_PP=$t timeout 600 python bench.py --config c5 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('c5 pp $t', r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['output_check']['all_finite'])"
done
