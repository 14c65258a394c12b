#!/bin/bash
# r05 session 7: kernel trace of the OPT stage alone (4 and 8 prompts) -> per-kernel / per-shape table
O=$PWD/gpurun_out/r05_s07; mkdir -p $O
R=$PWD
for P in 4 8; do
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof$P -o opt --output-format rocpd -- python $R/tools/opt_only.py $P 10 > $O/opt_only_$P.log 2>&1)
  tail -1 $O/opt_only_$P.log
  db=$(find $O/prof$P -name "*.db" | head -1)
  python tools/rocpd_summary.py $db $O/opt_kernels_$P.md --per-shape > /dev/null
  rm -rf $O/prof$P
done
timeout 300 python tools/opt_only.py 4 20 | tail -1
timeout 300 python tools/opt_only.py 8 20 | tail -1
timeout 600 python tools/r05_probe.py fp8lin > $O/fp8lin.log 2>&1; tail -12 $O/fp8lin.log
