#!/bin/bash
# r05 session 13: timing-only ablations of the STREAM64 loop (tools/_lib_s64ablN.so: bit 0 no activation pieces, 1 no weight pieces, 2 no fragment reads / MFMAs — after each
# tile's first K step; results wrong by construction) on the OPT stage alone
for n in 0 1 2 4 5 6 7; do
  lib=$PWD/tools/_lib_s64abl$n.so; [ $n = 0 ] && lib=$PWD/gill_amd/libgill_amd.so
  echo -n "ABL=$n: "; GILL_AMD_LIB=$lib timeout 300 python tools/opt_only.py 4 20 2>/dev/null | tail -1
done
