#!/bin/bash
one() { env "$@" timeout 600 python bench.py --prompts-per-gpu 8 --no-cpu-baseline --no-pmc --no-scale-origin --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$*: %.3f images/s, loop %.1f ms, frac %.4f' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))"; }
one GILL_X=0; one GILL_GEMM_PP128=0; one GILL_GEMM_MINSTEPS=12; one GILL_GEMM_MINSTEPS=48; one GILL_GEMM_PLAIN_PP=0; one GILL_UNET_XALG=0; one GILL_X=0
