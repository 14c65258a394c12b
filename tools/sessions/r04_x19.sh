#!/bin/bash
O=gpurun_out/r04_x19; mkdir -p $O
bash tools/prof.sh r04_x19/prof > $O/prof_head.txt 2>&1
grep -E "gemm_kernel<8, 128|gemm_kernel<4, 128, 1" $O/prof/kernel_stats.md | head -20
rm -rf $O/prof/prof
