#!/bin/bash
# cross-attention as two GEMMs (GILL_UNET_XALG): op-level test, engine switch test, oracle tests of the full-size engine, loop A/B
O=gpurun_out/r04_x1; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "cross_attention_folded" -s > $O/ops.log 2>&1; echo "ops rc=$?"; grep -E "xalg|passed|failed|Error|error" $O/ops.log | tail -n 30
timeout 1500 python -m pytest tests/test_configs_gpu.py -q -x -k "XALG" -s > $O/switch.log 2>&1; echo "switch rc=$?"; grep -E "full-size forward|passed|failed|Error" $O/switch.log | tail -n 8
one() { env "$@" timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$*: %.3f images/s, loop %.1f ms, frac %.4f, check %s' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r.get('output_check')))"; }
for r in 1 2; do one GILL_UNET_XALG=1; one GILL_UNET_XALG=0; done
