#!/bin/bash
# round 4, GPU session 2: weight prefetcher (A/B, parameter sweep, soak), write-through output stores (ops tests on that build, A/B)
set -x
O=gpurun_out/r04_s2; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -k "attention" -q -s > $O/attn_tests.log 2>&1; tail -3 $O/attn_tests.log
GILL_AMD_LIB=$PWD/tools/_lib_wt.so python -m pytest tests/test_ops_gpu.py -q -x > $O/ops_tests_wt.log 2>&1; tail -3 $O/ops_tests_wt.log
python -m pytest tests/test_soak_gpu.py -q -s > $O/soak.log 2>&1; tail -3 $O/soak.log
bash tools/ab_env.sh GILL_UNET_PREFETCH 3 > $O/ab_prefetch.log 2>&1; cat $O/ab_prefetch.log
bash tools/ab_bench.sh gill_amd/libgill_amd.so tools/_lib_wt.so 3 > $O/ab_wt.log 2>&1; cat $O/ab_wt.log
one() { env "$@" timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-scale-origin --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$*: %.3f images/s, loop %.1f ms, frac %.4f' % (r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))"; }
for r in 1 2; do
  one GILL_PF_SEG_MB=24; one GILL_PF_SEG_MB=96; one GILL_PF_BLOCKS=32; one GILL_PF_BLOCKS=128; one GILL_PF_BLOCKS=256; one GILL_PF_SEG_MB=48
done > $O/pf_sweep.log 2>&1; cat $O/pf_sweep.log
GILL_AMD_LIB=$PWD/tools/_lib_wt.so bash tools/ab_env.sh GILL_UNET_PREFETCH 1 > $O/ab_prefetch_on_wt.log 2>&1; cat $O/ab_prefetch_on_wt.log
ls -la $O
