set +x
# QKV on the ping-pong tiles, three-way on one box: previous build / all balanced shapes / only the 256-row (level-2) shapes
O=gpurun_out/r06_s26; mkdir -p $O
for r in 1 2 3 4; do for lib in tools/_lib_base.so gill_amd/libgill_amd.so tools/_lib_qkvl2.so; do
  GILL_AMD_LIB=$(realpath $lib) timeout 900 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$lib round $r: %.3f images/s, loop %.1f ms, frac %.4f, %.0f MHz, %.0f W' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r.get('sclk_mhz_mean') or 0, r.get('power_w_mean') or 0))"
done; done | tee $O/ab_loop.log
