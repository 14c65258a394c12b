#!/bin/bash
O=gpurun_out/r04_x11; mkdir -p $O
for v in 1 2; do GILL_GEMM_GEGLU_PP=$v GILL_GEMM_QKV_PP=$v timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "geglu or qkv or gemm" > $O/ops$v.log 2>&1; echo "ops v=$v rc=$?"; tail -n 1 $O/ops$v.log; done
one() { env "$@" timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$*: %.3f images/s, loop %.1f ms, frac %.4f' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))"; }
for r in 1 2; do one GILL_GEMM_GEGLU_PP=0; one GILL_GEMM_GEGLU_PP=1; one GILL_GEMM_GEGLU_PP=2; one GILL_GEMM_QKV_PP=1; one GILL_GEMM_QKV_PP=2; done
