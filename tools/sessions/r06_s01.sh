#!/bin/bash
# round 6, session 1: first run of the COOP finishes (gemm.hip EPI 6 / 7) — operator parity incl. bit-identity against the launches they replace,
# the stage / config tests that run full-size forwards, then the bench A/B (GILL_GEMM_COOP=0 vs 1) and a forward timeline
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s01; mkdir -p $O
export GILL_SKIP_SLOW=1
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -s -k "conv3x3 or gemm" > $O/ops.log 2>&1; echo "ops rc=$?" | tee -a $O/summary.txt
tail -5 $O/ops.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-pmc --no-scale-origin > $O/bench_quick.log 2> $O/bench_quick.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json
try:
  r=json.loads(open('gpurun_out/r06_s01/bench_quick.log').read().strip().splitlines()[-1])
  print('value %.3f frac %.4f loop %.1f ms fwdcheck %s clocks %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['forward_check'], r.get('clocks')))
except Exception as e: print('bench parse failed', e)
PY
tail -3 $O/bench_quick.err
bash tools/ab_env.sh GILL_GEMM_COOP 2 2>&1 | tee $O/ab_coop.log
timeout 900 bash tools/prof.sh r06_s01/prof > $O/prof_head.log 2>&1
db=$(find $O/prof -name "*.db" | head -1)
python tools/forward_timeline.py $db > $O/forward_timeline.txt 2>&1; head -3 $O/forward_timeline.txt; tail -2 $O/forward_timeline.txt
rm -rf $O/prof/prof
