#!/bin/bash
# r05 session 8: STREAM64 K-walk rotation (GILL_GEMM_KROT = 0 | 5 | 13) on the OPT stage alone; gemm operator tests on the rebuilt library
O=$PWD/gpurun_out/r05_s08; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "gemm" > $O/ops.log 2>&1; tail -2 $O/ops.log
for rep in 1 2; do for k in 0 5 13 29; do
  echo -n "KROT=$k rep $rep: "; GILL_GEMM_KROT=$k timeout 300 python tools/opt_only.py 4 20 2>/dev/null | tail -1
done; done
for k in 0 5; do echo -n "KROT=$k 8 prompts: "; GILL_GEMM_KROT=$k timeout 300 python tools/opt_only.py 8 20 2>/dev/null | tail -1; done
