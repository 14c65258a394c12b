#!/bin/bash
O=gpurun_out/r04_x30; mkdir -p $O
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_stages_gpu.py -q -x > $O/t.log 2>&1; echo "tests rc=$?"; tail -n 1 $O/t.log
python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('%.3f images/s, loop %.1f ms, frac %.4f' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac']), r['output_check'])"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 1
