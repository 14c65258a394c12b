#!/bin/bash
# r05 session 12: plain 128-row GEMMs on eight waves (GILL_GEMM_W8P=1): operator tests with the switch on, loop A/B
O=gpurun_out/r05_s12; mkdir -p $O
GILL_GEMM_W8P=1 timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "gemm or geglu or qkv or conv or groupnorm" > $O/ops_w8p.log 2>&1; tail -1 $O/ops_w8p.log
timeout 600 python -m pytest tests/test_stages_gpu.py -x -q -k "denoise_tiny" > $O/tiny.log 2>&1; tail -1 $O/tiny.log
bash tools/ab_env.sh GILL_GEMM_W8P 3
