#!/bin/bash
# round 4, GPU session 3: why does the prefetcher's side branch slow the captured loop?  graph vs eager vs packet capture off + a trace
set -x
O=gpurun_out/r04_s3; mkdir -p $O
one() { env "$@" timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$*: %.3f images/s, loop %.1f ms, frac %.4f' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))"; }
for r in 1 2; do
  one GILL_UNET_PREFETCH=0; one GILL_UNET_PREFETCH=1
  one GILL_NO_GRAPH=1 GILL_UNET_PREFETCH=0; one GILL_NO_GRAPH=1 GILL_UNET_PREFETCH=1
  one DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 GILL_UNET_PREFETCH=0; one DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 GILL_UNET_PREFETCH=1
  one GILL_NO_GRAPH=1 GILL_UNET_PREFETCH=1 GILL_PF_BLOCKS=16; one GILL_NO_GRAPH=1 GILL_UNET_PREFETCH=1 GILL_PF_BLOCKS=256
done > $O/matrix.log 2>&1; cat $O/matrix.log
GILL_UNET_PREFETCH=1 bash tools/prof.sh r04_s3/prof_pf > $O/prof_pf_head.txt 2>&1
db=$(find $O/prof_pf -name "*.db" | head -1); python tools/overlap_check.py $db > $O/overlap_graph.txt 2>&1; cat $O/overlap_graph.txt
GILL_NO_GRAPH=1 GILL_UNET_PREFETCH=1 bash tools/prof.sh r04_s3/prof_pf_eager > $O/prof_pf_eager_head.txt 2>&1
db=$(find $O/prof_pf_eager -name "*.db" | head -1); python tools/overlap_check.py $db > $O/overlap_eager.txt 2>&1; cat $O/overlap_eager.txt
rm -rf $O/prof_pf/prof $O/prof_pf_eager/prof
