#!/bin/bash
# round 4, GPU session 8: fused reducer widths
O=gpurun_out/r04_s8; mkdir -p $O
for w in 40 80; do GILL_RED_GN_W=$w python -m pytest tests/test_ops_gpu.py -k "fused_groupnorm" -q > $O/ops_w$w.log 2>&1; echo "W=$w: $(tail -n 1 $O/ops_w$w.log)"; done
one() { env "$@" timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$*: %.3f images/s, loop %.1f ms, frac %.4f' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))"; }
for r in 1 2 3; do one GILL_GEMM_RED_GN=0; one GILL_RED_GN_W=80; one GILL_RED_GN_W=40; one GILL_RED_GN_W=0; done > $O/matrix.log 2>&1; cat $O/matrix.log
bash tools/prof.sh r04_s8/prof > $O/prof_head.txt 2>&1
db=$(find $O/prof -name "*.db" | head -1); python tools/forward_timeline.py $db > $O/timeline.txt 2>&1; grep "reduce_gn" $O/timeline.txt | awk '{print $2,$3,$4,$5}' | sort | uniq -c
rm -rf $O/prof/prof
