#!/bin/bash
# round 6, session 7: fp8 GEGLU projections (linear_fp8.hip) — operator parity, the fp8 UNet mode against the fp32 oracle and the bf16 mode, then the
# configs[4] bench with and without them (16 prompts per GPU, alternating)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s07; mkdir -p $O
timeout 900 python -m pytest tests/test_fp8_gpu.py -x -q -s > $O/fp8_tests.log 2>&1; echo "fp8 tests rc=$?" | tee $O/summary.txt
grep -E "^\[|vs un-quantised|passed|failed" $O/fp8_tests.log | tail -30
for r in 1 2; do
  for mode in "--fp8-convs-only" ""; do
    timeout 900 python bench.py --config c5 $mode --steps 4 --warmup 2 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('c5 $mode round $r: %.3f images/s, loop %.1f ms (16 prompts), frac %.4f, sclk %s' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r.get('sclk_mhz_mean')))"
  done
done 2>&1 | tee $O/ab_c5.log
timeout 900 python bench.py --prompts-per-gpu 16 --steps 4 --warmup 2 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('bf16 16 prompts: %.3f images/s, loop %.1f ms, frac %.4f' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))" | tee -a $O/ab_c5.log
