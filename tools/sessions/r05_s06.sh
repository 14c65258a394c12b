#!/bin/bash
# r05 session 6: STREAM64 with 64 x 64-blocked weights + nt loads, reduce + residual + LayerNorm: operator test, OPT tests (incl. full depth), bench stage A/B
O=gpurun_out/r05_s06; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -s -k "gemm" > $O/ops.log 2>&1; tail -4 $O/ops.log
timeout 1500 python -m pytest tests/test_stages_gpu.py tests/test_coverage_gpu.py -x -q -s -k "opt or log_likelihood or generate or gillmodel or public_api or kv_cache" > $O/tests.log 2>&1; grep -E "FULL DEPTH|passed|failed|Error|log-likelihood" $O/tests.log | tail -8
for rep in 1 2; do for v in "0 0" "1 0" "1 1"; do set -- $v
  GILL_GEMM_STREAM64=$1 GILL_OPT_REDUCE_LN=$2 timeout 900 python bench.py --steps 4 --warmup 2 --no-pmc --no-cpu-baseline > $O/bench_s$1_r$2_$rep.log 2>&1
  python3 - <<PY
import json
for l in open("$O/bench_s$1_r$2_$rep.log"):
    if l.startswith("{"):
        d = json.loads(l); print("STREAM64=$1 REDUCE_LN=$2 rep $rep: opt ms", round(d["stages"]["opt"]["ms"], 3), "TB/s", round(d["stages"]["opt"]["TBps"], 2), "img/s", round(d["value"], 3), "loop ms", round(d["roofline"]["avg_launch_ms"], 1), "origin img/s", round(d["scale_origin"]["value"], 3))
PY
done; done
