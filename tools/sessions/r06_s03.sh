#!/bin/bash
# round 6, session 3: the persistent ResnetBlock2D prototype (tools/ubench/persist_resnet.hip) at levels 3 and 2; ops parity after the gemm_tile
# refactor; loop A/B of GILL_GEMM_COOP = 0 / 1 (default: EPI 6 only) / 2
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s03; mkdir -p $O
export GILL_SKIP_SLOW=1
timeout 300 ./tools/ubench/persist_resnet 3 2>&1 | tee $O/persist_level3.log
timeout 300 ./tools/ubench/persist_resnet 2 2>&1 | tee $O/persist_level2.log
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "conv3x3 or gemm" > $O/ops.log 2>&1; echo "ops rc=$?" | tee -a $O/summary.txt; tail -2 $O/ops.log
for r in 1 2; do for v in 0 1 2; do
  GILL_GEMM_COOP=$v timeout 900 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 6 --warmup 2 2>/dev/null | tail -1 | \
    python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('GILL_GEMM_COOP=$v round $r: %.3f images/s, loop %.1f ms, frac %.4f, sclk %s W %s' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r.get('sclk_mhz_mean'), r.get('power_w_mean')))"
done; done 2>&1 | tee $O/ab_coop3.log
