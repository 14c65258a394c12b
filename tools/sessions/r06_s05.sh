#!/bin/bash
# round 6, session 5: the whole GPU suite on the round-6 build (COOP EPI 6 default, block map, gemm_tile refactor), with stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s05; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -s -x > $O/gpu_tests.log 2>&1; echo "suite rc=$?" | tee $O/summary.txt
tail -5 $O/gpu_tests.log
grep -n "FAILED\|Error" $O/gpu_tests.log | head -20
