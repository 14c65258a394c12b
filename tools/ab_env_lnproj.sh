# A/B via env switch: GILL_UNET_LNPROJ=0 vs default, alternating
for r in 1 2 3; do
  for v in 0 1; do
    GILL_UNET_LNPROJ=$v timeout 900 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 6 --warmup 2 2>/dev/null | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('lnproj=$v round $r: %.3f images/s, loop %.1f ms, frac %.4f, check %s' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r.get('forward_check')))"
  done
done
