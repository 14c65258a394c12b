set +x
# two probes on top of the K >= 640 ping-pong rule: (a) 64 x 160 tiles where they are one balanced round (2048 x 1280 x 1280: 256 instead of 320 workgroups), (b) per-sample weights on the ping-pong tiles
O=gpurun_out/r06_s22; mkdir -p $O
for lib in gill_amd/libgill_amd.so tools/_lib_tw160.so; do echo "== $lib"; GILL_AMD_LIB=$(realpath $lib) python tools/pp_shortk_probe.py 2>&1 | grep "2048 x 1280 x 1280\|512 x 1280"; done | tee $O/probe.log
for r in 1 2 3; do for lib in gill_amd/libgill_amd.so tools/_lib_tw160.so tools/_lib_ppwb.so; do
  GILL_AMD_LIB=$(realpath $lib) timeout 900 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$lib round $r: %.3f images/s, loop %.1f ms, finite %s' % (r['value'], r['roofline']['avg_launch_ms'], r['output_check']['all_finite']))"
done; done | tee $O/ab_loop.log
