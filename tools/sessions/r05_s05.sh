#!/bin/bash
# r05 session 5: STREAM64 (128 x 64 tile, 6-deep ring) for the OPT weight-streaming GEMMs: cold-weight operator probe A/B, OPT tests, bench stage A/B
O=gpurun_out/r05_s05; mkdir -p $O
for v in 0 1; do
  GILL_GEMM_STREAM64=$v timeout 600 python tools/r05_probe.py optcold > $O/optcold_stream$v.log 2>&1; echo "STREAM64=$v"; tail -17 $O/optcold_stream$v.log
done
timeout 1200 python -m pytest tests/test_stages_gpu.py tests/test_coverage_gpu.py -x -q -s -k "opt or log_likelihood or generate or gillmodel or public_api" > $O/tests.log 2>&1; tail -5 $O/tests.log
for rep in 1 2; do for v in 0 1; do
  GILL_GEMM_STREAM64=$v timeout 900 python bench.py --steps 4 --warmup 2 --no-pmc --no-scale-origin --no-cpu-baseline > $O/bench_v${v}_$rep.log 2>&1
  python3 - <<PY
import json
for l in open("$O/bench_v${v}_$rep.log"):
    if l.startswith("{"):
        d = json.loads(l); print("STREAM64=$v rep $rep: opt ms", round(d["stages"]["opt"]["ms"], 3), "TB/s", round(d["stages"]["opt"]["TBps"], 2), "img/s", round(d["value"], 3), "loop ms", round(d["roofline"]["avg_launch_ms"], 1))
PY
done; done
