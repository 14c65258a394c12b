#!/bin/bash
# r05 session 9: STREAM64 on eight waves of 32 x 32 (GILL_GEMM_S64W=8) vs four of 64 x 32 (=4): gemm operator tests, OPT stage alone, OPT parity tests
O=$PWD/gpurun_out/r05_s09; mkdir -p $O
for w in 4 8; do GILL_GEMM_S64W=$w timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "gemm" > $O/ops_w$w.log 2>&1; tail -1 $O/ops_w$w.log; done
for rep in 1 2; do for w in 4 8; do
  echo -n "S64W=$w rep $rep: "; GILL_GEMM_S64W=$w timeout 300 python tools/opt_only.py 4 20 2>/dev/null | tail -1
done; done
for w in 4 8; do echo -n "S64W=$w 8 prompts: "; GILL_GEMM_S64W=$w timeout 300 python tools/opt_only.py 8 20 2>/dev/null | tail -1; done
timeout 1500 python -m pytest tests/test_stages_gpu.py tests/test_coverage_gpu.py -x -q -s -k "opt or log_likelihood or generate or gillmodel or kv_cache" > $O/tests.log 2>&1; grep -E "FULL DEPTH|passed|failed|Error" $O/tests.log | tail -5
