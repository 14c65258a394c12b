#!/bin/bash
# round 4, GPU session 5: runtime knobs for the launch path (kernarg placement, eager vs graph)
O=gpurun_out/r04_s5; mkdir -p $O
one() { env "$@" timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$*: %.3f images/s, loop %.1f ms, frac %.4f, step %.1f ms' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['ms_per_step']))"; }
for r in 1 2 3; do
  one GILL_ATT_DMA=1; one GILL_ATT_DMA=1 HIP_FORCE_DEV_KERNARG=1; one GILL_ATT_DMA=1 GILL_NO_GRAPH=1; one GILL_ATT_DMA=1 GILL_NO_GRAPH=1 HIP_FORCE_DEV_KERNARG=1
done > $O/matrix.log 2>&1; cat $O/matrix.log
