#!/bin/bash
# r05 session 3: Winograd level-2 convolutions — operator parity, operator timing, loop A/B (GILL_CONV_WINO=0 / 1)
R=$PWD; O=$R/gpurun_out/r05_s03; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -s -k "winograd or fused_groupnorm" > $O/ops.log 2>&1; tail -15 $O/ops.log
timeout 600 python tools/r05_probe.py winoop > $O/winoop.log 2>&1; tail -6 $O/winoop.log
for rep in 1 2; do for wv in 0 1; do
  GILL_CONV_WINO=$wv timeout 900 python bench.py --steps 6 --warmup 2 --no-pmc --no-scale-origin > $O/bench_w${wv}_$rep.log 2>&1
  python3 - <<PY
import json
for l in open("$O/bench_w${wv}_$rep.log"):
    if l.startswith("{"):
        d = json.loads(l); print("WINO=$wv rep $rep: loop ms", round(d["roofline"]["avg_launch_ms"], 2), "img/s", round(d["value"], 3), "frac", round(d["roofline"]["frac"], 4), "opt ms", round(d["stages"]["opt"]["ms"], 2), "forward_check", d.get("forward_check"))
PY
done; done
