#!/bin/bash
O=gpurun_out/r04_b8b; mkdir -p $O
for i in 1 2 3 4 5; do python bench.py --gpus 8 --backend gloo --share-gpu --small --total-prompts 61 --infer-steps 2 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $O/base.$i.log 2>&1; echo "base run $i rc=$? $(grep -c 'BAD' $O/base.$i.log) bad lines"; done
for i in 1 2 3; do GILL_NO_GRAPH=1 python bench.py --gpus 8 --backend gloo --share-gpu --small --total-prompts 61 --infer-steps 2 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $O/ng.$i.log 2>&1; echo "nograph run $i rc=$? $(grep -c 'BAD' $O/ng.$i.log) bad lines"; done
python -m pytest tests/test_configs_gpu.py -k "eight_ranks or two_ranks" -q 2>&1 | tail -n 2
one() { env "$@" timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$*: %.3f images/s, loop %.1f ms, frac %.4f' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))"; }
for r in 1 2; do one GILL_ATT_DMA=1; one GILL_ATT_DMA=0; done
