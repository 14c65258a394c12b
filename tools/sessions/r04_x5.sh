#!/bin/bash
O=gpurun_out/r04_x5; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "gemm" -s > $O/ops.log 2>&1; echo "ops rc=$?"; tail -n 1 $O/ops.log; grep "long-K" $O/ops.log
bash tools/ab_env.sh GILL_GEMM_PLAIN_PP 3 > $O/ab.log 2>&1; cat $O/ab.log
