#!/bin/bash
# round 4, GPU session 1: boundary micro-benchmark, LDS-DMA attention (tests, kernel time, loop A/B), weight-prefetch potential
set -x
O=gpurun_out/r04_s1; mkdir -p $O
./tools/ubench/boundary > $O/boundary.log 2>&1; tail -8 $O/boundary.log
python -m pytest tests/test_ops_gpu.py -k "attention" -q -s > $O/attn_tests.log 2>&1; tail -5 $O/attn_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
# kernel time of the level-0 self-attention, both kernels, from a kernel trace
for v in 0 1; do
  (cd /tmp && export TMPDIR=/tmp && GILL_ATT_DMA=$v GILL_OP_REPEAT=20 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/attn_prof$v -o a --output-format csv -- python $OLDPWD/tools/one_op.py attn 8 8 4096 4096 40 > $OLDPWD/$O/attn_op$v.log 2>&1)
  f=$(find $O/attn_prof$v -name "*kernel_stats.csv" | head -1); grep -i "attention" $f | cut -c1-200
  tail -1 $O/attn_op$v.log
done
rm -rf $O/attn_prof0 $O/attn_prof1
python -m pytest tests/test_configs_gpu.py -k "switches" -q -s > $O/switch_tests.log 2>&1; tail -5 $O/switch_tests.log
bash tools/ab_env.sh GILL_ATT_DMA 2 > $O/ab_att_dma.log 2>&1; cat $O/ab_att_dma.log
# weight-prefetch potential: per-kernel durations with every GEMM's weights touched right before it
bash tools/prof.sh r04_s1/prof_base > $O/prof_base_head.txt 2>&1
db=$(find $O/prof_base -name "*.db" | head -1); python tools/forward_timeline.py $db > $O/timeline_base.txt 2>&1; head -2 $O/timeline_base.txt; tail -1 $O/timeline_base.txt
GILL_UNET_TOUCH_W=1 bash tools/prof.sh r04_s1/prof_touch > $O/prof_touch_head.txt 2>&1
db=$(find $O/prof_touch -name "*.db" | head -1); python tools/forward_timeline.py $db > $O/timeline_touch.txt 2>&1; head -2 $O/timeline_touch.txt; tail -1 $O/timeline_touch.txt
rm -rf $O/prof_base/prof $O/prof_touch/prof
ls -la $O
