#!/bin/bash
one() { env "$@" timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$*: %.3f images/s, opt %.2f ms (%.2f TB/s), loop %.1f ms' % (r['value'], r['stages']['opt']['ms'], r['stages']['opt']['TBps'], r['roofline']['avg_launch_ms']))"; }
for v in 4 8 16 32; do one GILL_GEMM_SKINNY_MINSTEPS=$v; done
one GILL_GEMM_BM=64
