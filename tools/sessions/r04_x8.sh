#!/bin/bash
O=gpurun_out/r04_x8; mkdir -p $O
GILL_GEMM_T64=7 timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "gemm or geglu or qkv" > $O/ops.log 2>&1; echo "ops rc=$?"; tail -n 1 $O/ops.log
one() { env "$@" timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$*: %.3f images/s, loop %.1f ms, frac %.4f' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))"; }
for r in 1 2; do for m in 0 1 2 4; do one GILL_GEMM_T64=$m; done; done
