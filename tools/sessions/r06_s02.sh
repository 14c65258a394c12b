#!/bin/bash
# round 6, session 2: the COOP finishes after the all-loads-up-front fix, and the XCD block map (GILL_XMAP=0: the range maps), operator level first,
# then the loop (A/B of both switches) and the fetch-by-kernel table
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s02; mkdir -p $O
export GILL_SKIP_SLOW=1
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "conv3x3 or gemm" > $O/ops.log 2>&1; echo "ops rc=$?" | tee -a $O/summary.txt; tail -2 $O/ops.log
timeout 300 python tools/coop_bench.py 2>&1 | grep -v Warning | tee $O/coop_bench_xmap1.log
GILL_XMAP=0 timeout 300 python tools/coop_bench.py 2>&1 | grep -v Warning | tee $O/coop_bench_xmap0.log
bash tools/ab_env.sh GILL_GEMM_COOP 2 2>&1 | tee $O/ab_coop.log
bash tools/ab_env.sh GILL_XMAP 2 2>&1 | tee $O/ab_xmap.log
timeout 600 python tools/pmc_by_kernel.py --out $O/fetch_by_kernel.md > $O/pmc.log 2>&1; head -12 $O/fetch_by_kernel.md
