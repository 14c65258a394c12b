#!/bin/bash
# timing-only ablations of the ping-pong conv loop (tools/_lib_abl*.so built with -DPP_ABL=mask: 1 no fragment reads, 2 no LDS-DMA, 4 no MFMAs,
# 8 no pointer bookkeeping, after the first K step): level-0 (256 x 160 tiles) and level-1 (128 x 160 tiles) convolutions
R=$PWD; O=$R/gpurun_out/r04_x27; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in gill_amd/libgill_amd.so tools/_lib_abl1.so tools/_lib_abl2.so tools/_lib_abl4.so tools/_lib_abl8.so tools/_lib_abl3.so tools/_lib_abl7.so tools/_lib_abl15.so; do
  for shape in "8 64 64 320 0 320" "8 32 32 640 0 640"; do
    rm -rf $O/p
    GILL_AMD_LIB=$R/$lib GILL_OP_REPEAT=30 rocprofv3 --kernel-trace --stats -d $O/p -o x --output-format csv -- python $R/tools/one_op.py conv $shape > $O/run.log 2>&1
    f=$(find $O/p -name "*kernel_stats.csv" | head -1)
    echo "$lib [$shape]: $(grep 'gemm_kernel<8' $f | head -1 | awk -F'","|",' '{print $1}' | sed 's/.*gemm_kernel/gemm_kernel/; s/(.*//') avg_ns=$(grep 'gemm_kernel<8' $f | head -1 | awk -F, '{print $(NF-4)}')"
  done
done
rm -rf $O/p
