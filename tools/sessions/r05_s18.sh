#!/bin/bash
# r05 session 18: where a GEMM launch's fixed cost goes — timing-only builds of the STREAM64 kernels with an empty main loop (31) and, on top, no epilogue (+64),
# no prologue loads / first step (+128), neither (+192), or an immediate return (32): per-kernel durations from a kernel trace of the OPT stage
O=$PWD/gpurun_out/r05_s18; mkdir -p $O
R=$PWD
for n in 31 95 159 223 32; do
  lib=$R/tools/_lib_s64abl$n.so
  (cd /tmp && export TMPDIR=/tmp && GILL_AMD_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof$n -o opt --output-format rocpd -- python $R/tools/opt_only.py 4 10 > $O/opt_only_$n.log 2>&1)
  echo "ABL=$n: $(tail -1 $O/opt_only_$n.log)"
  db=$(find $O/prof$n -name "*.db" | head -1)
  python tools/rocpd_summary.py $db $O/opt_kernels_abl$n.md --per-shape > /dev/null
  grep -E "^\| .gemm_kernel<8, 64.*\| 832 \||^\| .gemm_kernel<8, 64.*\| 416 \||reduce_ln|attention_kernel" $O/opt_kernels_abl$n.md | head -5 | cut -c1-90
  rm -rf $O/prof$n
done
