#!/bin/bash
# A/B of one engine switch (an environment variable read once per process), alternating runs of the default bench workload on one box:
#   tools/ab_env.sh GILL_UNET_LNPROJ [rounds]        -> VAR=0 vs VAR=1
VAR=$1; R=${2:-3}
for r in $(seq 1 $R); do
  for v in 0 1; do
    env $VAR=$v timeout 900 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 6 --warmup 2 2>/dev/null | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$VAR=$v round $r: %.3f images/s, loop %.1f ms, frac %.4f' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))"
  done
done
