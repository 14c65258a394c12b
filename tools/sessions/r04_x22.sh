#!/bin/bash
one() { env "$@" timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$*: %.3f images/s, loop %.1f ms, frac %.4f' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))"; }
for r in 1 2; do one GILL_GN_MINBLOCKS=1024; one GILL_GN_MINBLOCKS=512; one GILL_GN_MINBLOCKS=256; one GILL_GN_MINBLOCKS=2048; one GILL_GN_ROWS=128 GILL_GN_MINBLOCKS=512; done
