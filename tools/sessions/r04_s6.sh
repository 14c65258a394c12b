#!/bin/bash
# round 4, GPU session 6: interleaved attention kernel (dma3) — tests per mode, kernel time, loop
O=gpurun_out/r04_s6; mkdir -p $O
for m in 2 3 4; do GILL_ATT_DMA=$m python -m pytest tests/test_ops_gpu.py -k "attention" -q > $O/attn_tests_m$m.log 2>&1; echo "mode $m: $(tail -n 1 $O/attn_tests_m$m.log)"; done
for v in 1 2 3 4; do
  (cd /tmp && export TMPDIR=/tmp && GILL_ATT_DMA=$v GILL_OP_REPEAT=20 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/p$v -o a --output-format csv -- python $OLDPWD/tools/one_op.py attn 8 8 4096 4096 40 > $OLDPWD/$O/op$v.log 2>&1)
  f=$(find $O/p$v -name "*kernel_stats.csv" | head -1); echo "mode $v: $(grep -i 'attention_dma' $f | cut -d, -f1-4)"; rm -rf $O/p$v
done
one() { env "$@" timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$*: %.3f images/s, loop %.1f ms, frac %.4f' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))"; }
for r in 1 2; do one GILL_ATT_DMA=1; one GILL_ATT_DMA=2; one GILL_ATT_DMA=3; one GILL_ATT_DMA=4; done > $O/matrix.log 2>&1; cat $O/matrix.log
