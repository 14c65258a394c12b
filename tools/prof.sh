#!/bin/bash
# rocprofv3 kernel trace of the bench command -> per-kernel + per-shape summary under gpurun_out/<tag>/ (copy into profiles/)
#   tools/prof.sh <tag> [extra bench args]
tag=$1; shift
out=$PWD/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/prof -o bench --output-format rocpd -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-scale-origin "$@" > $out/bench_under_prof.log 2>&1
cd $OLDPWD
db=$(find $out/prof -name "*.db" | head -1)
python tools/rocpd_summary.py $db $out/kernel_stats.md --per-shape > /dev/null
head -45 $out/kernel_stats.md
