#!/bin/bash
O=gpurun_out/r04_x13; mkdir -p $O
one() { env "$@" timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$*: %.3f images/s, loop %.1f ms, frac %.4f' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))"; }
for r in 1 2; do one GILL_X=0; one GILL_CONV_KORDER=1; one GILL_CONV_KORDER=0; done
python - <<'PY'
import json, subprocess, sys
r = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-pmc", "--no-scale-origin", "--steps", "2", "--warmup", "1"], capture_output=True, text=True)
rec = json.loads(r.stdout.strip().splitlines()[-1])
for k in rec["roofline_kernels"]: print(k["kernel"][:60], "%.1f us" % k["avg_launch_us"], "frac %.3f" % k["frac"])
PY
