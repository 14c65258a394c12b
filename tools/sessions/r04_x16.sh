#!/bin/bash
O=gpurun_out/r04_x16; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "attention" > $O/ops.log 2>&1; echo "ops rc=$?"; tail -n 3 $O/ops.log
python - <<'PY'
import json, subprocess, sys
r = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-pmc", "--no-scale-origin", "--steps", "4", "--warmup", "2"], capture_output=True, text=True)
rec = json.loads(r.stdout.strip().splitlines()[-1])
print("%.3f images/s loop %.1f ms frac %.4f" % (rec["value"], rec["roofline"]["avg_launch_ms"], rec["roofline"]["frac"]), rec["output_check"])
for k in rec["roofline_kernels"]: print(k["kernel"][:60], "%.1f us" % k["avg_launch_us"], "frac %.3f" % k["frac"])
PY
