#!/bin/bash
O=$PWD/gpurun_out/r04_x3; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in 0 2; do
  GILL_XALG_TILE=$v GILL_OP_REPEAT=20 rocprofv3 --kernel-trace --stats -d $O/p$v -o x --output-format csv -- python $R/tools/xalg_bench.py > $O/run$v.log 2>&1
  f=$(find $O/p$v -name "*kernel_stats.csv" | head -1); echo "tile $v"; grep "gemm_kernel" $f | sed 's/(GemmDev)//' | awk -F'","' '{print $1, $2, $3, $4}' | cut -c1-160 | head -12
  cp $f $O/stats$v.csv; rm -rf $O/p$v
done
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "cross_attention_folded" > $O/ops.log 2>&1; tail -n 1 $O/ops.log
