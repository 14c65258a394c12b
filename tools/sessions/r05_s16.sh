#!/bin/bash
# r05 session 16: finer timing-only ablations of the STREAM64 loop (bits: 1 no activation pieces, 2 no weight pieces, 4 no fragment reads / MFMAs, 8 no barrier, 16 no issue() at all)
# -> per-kernel durations of the OPT stage from a kernel trace
O=$PWD/gpurun_out/r05_s16; mkdir -p $O
R=$PWD
for n in 0 3 5 7 15 23 31; do
  lib=$R/tools/_lib_s64abl$n.so; [ $n = 0 ] && lib=$R/gill_amd/libgill_amd.so
  (cd /tmp && export TMPDIR=/tmp && GILL_AMD_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof$n -o opt --output-format rocpd -- python $R/tools/opt_only.py 4 10 > $O/opt_only_$n.log 2>&1)
  echo "ABL=$n: $(tail -1 $O/opt_only_$n.log)"
  db=$(find $O/prof$n -name "*.db" | head -1)
  python tools/rocpd_summary.py $db $O/opt_kernels_abl$n.md --per-shape > /dev/null
  grep -E "^\| .gemm_kernel<8, 64.*\| 832 \||^\| .gemm_kernel<8, 64.*\| 416 \|" $O/opt_kernels_abl$n.md | head -3 | cut -c1-90
  rm -rf $O/prof$n
done
