#!/bin/bash
# more timing-only ablations of the ping-pong conv kernel: bit 4 (16) no epilogue, bit 5 (32) no barriers in the loop
R=$PWD; O=$R/gpurun_out/r04_x29; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in gill_amd/libgill_amd.so tools/_lib_abl16.so tools/_lib_abl15.so tools/_lib_abl31.so tools/_lib_abl47.so tools/_lib_abl63.so; do
  for shape in "8 64 64 320 0 320" "8 32 32 640 0 640"; do
    rm -rf $O/p
    GILL_AMD_LIB=$R/$lib GILL_OP_REPEAT=30 rocprofv3 --kernel-trace --stats -d $O/p -o x --output-format csv -- python $R/tools/one_op.py conv $shape > $O/run.log 2>&1
    f=$(find $O/p -name "*kernel_stats.csv" | head -1)
    echo "$lib [$shape]: avg_ns=$(grep 'gemm_kernel<8' $f | head -1 | awk -F, '{print $(NF-4)}')"
  done
done
rm -rf $O/p
