#!/bin/bash
# round 6, session 6: the granule hand-off (R2) of the in-epilogue GroupNorm finish — operator parity, per-shape timing, loop A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s06; mkdir -p $O
export GILL_SKIP_SLOW=1
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "conv3x3 or gemm" > $O/ops.log 2>&1; echo "ops rc=$?" | tee -a $O/summary.txt; tail -2 $O/ops.log
timeout 300 python tools/coop_bench.py 2>&1 | grep -v Warning | tee $O/coop_bench.log
bash tools/ab_env.sh GILL_GEMM_COOP 3 2>&1 | tee $O/ab_coop.log
timeout 600 python -m pytest tests/test_configs_gpu.py -x -q -k "switches" > $O/switches.log 2>&1; echo "switches rc=$?" | tee -a $O/summary.txt; tail -2 $O/switches.log
