#!/bin/bash
# r05 session 1: wave-specialised conv main loop prototype (tools/ubench/gemm_ws.hip) + its LDS bank-conflict counters
R=$PWD; O=$R/gpurun_out/r05_s01; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 $R/tools/ubench/gemm_ws > $O/gemm_ws.log 2>&1
tail -30 $O/gemm_ws.log
# (PMC pass: see the first run of this session)
