#!/bin/bash
O=gpurun_out/r04_x20; mkdir -p $O
GILL_PP128_MINROWS=1024 timeout 900 python -m pytest tests/test_stages_gpu.py -q -k "unet_forward_sd15_vs_oracle" -s > $O/t.log 2>&1; echo "test rc=$?"; grep -E "rel|passed|failed" $O/t.log | tail -n 3
one() { env "$@" timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$*: %.3f images/s, loop %.1f ms, frac %.4f' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))"; }
for r in 1 2 3; do one GILL_PP128_MINROWS=0; one GILL_PP128_MINROWS=1024; one GILL_PP128_MINROWS=4096; done
