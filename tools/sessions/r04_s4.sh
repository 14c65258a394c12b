#!/bin/bash
# round 4, GPU session 4: pipelined attention kernel variants (tests + kernel time + loop), packet-capture default
O=gpurun_out/r04_s4; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -k "attention" -q -s > $O/attn_tests.log 2>&1; tail -n 3 $O/attn_tests.log
GILL_ATT_DMA=3 python -m pytest tests/test_ops_gpu.py -k "attention" -q > $O/attn_tests_m3.log 2>&1; tail -n 2 $O/attn_tests_m3.log
for v in 1 2 3; do
  (cd /tmp && export TMPDIR=/tmp && GILL_ATT_DMA=$v GILL_OP_REPEAT=20 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/attn_prof$v -o a --output-format csv -- python $OLDPWD/tools/one_op.py attn 8 8 4096 4096 40 > $OLDPWD/$O/attn_op$v.log 2>&1)
  f=$(find $O/attn_prof$v -name "*kernel_stats.csv" | head -1); grep -i "attention" $f | cut -c1-120
done
rm -rf $O/attn_prof1 $O/attn_prof2 $O/attn_prof3
one() { env "$@" timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$*: %.3f images/s, loop %.1f ms, frac %.4f' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))"; }
for r in 1 2 3; do
  one GILL_ATT_DMA=1; one GILL_ATT_DMA=2; one GILL_ATT_DMA=3; one GILL_ATT_DMA=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
done > $O/matrix.log 2>&1; cat $O/matrix.log
