#!/bin/bash
# r05 session 11: GEGLU / QKV 128 x 128 tiles on eight waves (GILL_GEMM_W8=1): operator tests with the switch on, operator timings, loop A/B
O=gpurun_out/r05_s11; mkdir -p $O
GILL_GEMM_W8=1 timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "geglu or qkv or gemm" > $O/ops_w8.log 2>&1; tail -1 $O/ops_w8.log
for v in 0 1; do GILL_GEMM_W8=$v timeout 600 python tools/r05_probe.py fp8lin 2>/dev/null | grep -E "level|mid" | head -3 | sed "s/^/W8=$v /"; done
bash tools/ab_env.sh GILL_GEMM_W8 3
