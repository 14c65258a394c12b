#!/bin/bash
O=gpurun_out/r04_x18; mkdir -p $O
timeout 900 python -m pytest tests/test_stages_gpu.py tests/test_ops_gpu.py -q -k "vae or conv" > $O/t.log 2>&1; echo "tests rc=$?"; tail -n 2 $O/t.log
one() { env "$@" timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$*: %.3f images/s, vae %.2f ms, loop %.1f ms' % (r['value'], r['stages']['vae']['ms'], r['roofline']['avg_launch_ms']))"; }
for r in 1 2 3; do one GILL_GEMM_PP_W128=0; one GILL_GEMM_PP_W128=1; done
