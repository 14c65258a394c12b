#!/bin/bash
one() { GILL_AMD_LIB=$(realpath $1) timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$1: %.3f images/s, loop %.1f ms, frac %.4f, attn %.1f us' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['roofline_kernels'][1]['avg_launch_us']))"; }
for r in 1 2 3; do one tools/_lib_base.so; one gill_amd/libgill_amd.so; one tools/_lib_prio.so; done
