#!/bin/bash
# round 6, session 4: prototype rerun (variant A fixed), EPI 6 after the reorder (arrive first, raw stores after, gamma / beta prefetched)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s04; mkdir -p $O
export GILL_SKIP_SLOW=1
timeout 300 ./tools/ubench/persist_resnet 3 2>&1 | tee $O/persist_level3.log
timeout 300 ./tools/ubench/persist_resnet 2 2>&1 | tee $O/persist_level2.log
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "conv3x3 or gemm" > $O/ops.log 2>&1; echo "ops rc=$?" | tee -a $O/summary.txt; tail -2 $O/ops.log
timeout 300 python tools/coop_bench.py 2>&1 | grep -v Warning | tee $O/coop_bench.log
bash tools/ab_env.sh GILL_GEMM_COOP 3 2>&1 | tee $O/ab_coop.log
