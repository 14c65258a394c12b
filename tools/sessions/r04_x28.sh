#!/bin/bash
O=gpurun_out/r04_x28; mkdir -p $O
GILL_AMD_LIB=$PWD/tools/_lib_dmafirst.so timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "conv or gemm" > $O/ops.log 2>&1; echo "ops rc=$?"; tail -n 1 $O/ops.log
bash tools/ab_bench.sh gill_amd/libgill_amd.so tools/_lib_dmafirst.so 3 > $O/ab.log 2>&1; cat $O/ab.log
