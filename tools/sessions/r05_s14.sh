#!/bin/bash
# r05 session 14: kernel traces of the OPT stage with the shipped library and with the empty-loop ablation build (S64_ABL=7): per-kernel fixed cost
O=$PWD/gpurun_out/r05_s14; mkdir -p $O
R=$PWD
for n in 0 7; do
  lib=$R/tools/_lib_s64abl$n.so; [ $n = 0 ] && lib=$R/gill_amd/libgill_amd.so
  (cd /tmp && export TMPDIR=/tmp && GILL_AMD_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof$n -o opt --output-format rocpd -- python $R/tools/opt_only.py 4 10 > $O/opt_only_$n.log 2>&1)
  tail -1 $O/opt_only_$n.log
  db=$(find $O/prof$n -name "*.db" | head -1)
  python tools/rocpd_summary.py $db $O/opt_kernels_abl$n.md --per-shape > /dev/null
  grep -E "gemm_kernel|reduce_ln|attention_kernel" $O/opt_kernels_abl$n.md | head -12
  rm -rf $O/prof$n
done
