#!/bin/bash
O=gpurun_out/r04_x28; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "conv or gemm" > $O/ops.log 2>&1; echo "ops rc=$?"; tail -n 1 $O/ops.log
bash tools/ab_bench.sh tools/_lib_base.so gill_amd/libgill_amd.so 3 > $O/ab.log 2>&1; cat $O/ab.log
