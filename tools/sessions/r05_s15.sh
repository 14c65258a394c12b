#!/bin/bash
# r05 session 15: STREAM64 loop with immediate vmcnt waits: gemm operator tests, OPT stage alone (4 / 8 prompts), kernel trace
O=$PWD/gpurun_out/r05_s15; mkdir -p $O
R=$PWD
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "gemm" > $O/ops.log 2>&1; tail -1 $O/ops.log
for P in 4 4 8; do timeout 300 python tools/opt_only.py $P 20 2>/dev/null | tail -1; done
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o opt --output-format rocpd -- python $R/tools/opt_only.py 4 10 > $O/opt_only.log 2>&1)
db=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_summary.py $db $O/opt_kernels.md --per-shape > /dev/null
grep -E "gemm_kernel|reduce_ln|attention_kernel" $O/opt_kernels.md | head -12
rm -rf $O/prof
timeout 1500 python -m pytest tests/test_stages_gpu.py tests/test_coverage_gpu.py -x -q -k "opt or log_likelihood or generate or gillmodel or kv_cache" > $O/tests.log 2>&1; tail -1 $O/tests.log
