#!/bin/bash
# A/B of two builds of libgill_amd in alternating runs of the default bench workload (same box, same process sequence):
#   tools/ab_bench.sh <libA.so> <libB.so> [rounds] [extra bench args]
# prints images/s, the UNet loop's HIP-event time, the roofline fraction and the clock / power means of every run.
A=$1; B=$2; R=${3:-2}; shift 3
for r in $(seq 1 $R); do
  for lib in "$A" "$B"; do
    GILL_AMD_LIB=$(realpath $lib) timeout 900 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 6 --warmup 2 "$@" 2>/dev/null | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$lib round $r: %.3f images/s, loop %.1f ms, frac %.4f, %.0f MHz, %.0f W' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r.get('sclk_mhz_mean') or 0, r.get('power_w_mean') or 0))"
  done
done
