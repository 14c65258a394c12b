set +x
# round 6 evidence batch on the current build: GPU suite with stats, driver bench command, rocprofv3 kernel stats + forward timeline, MFMA-busy and fetch PMC passes, C4 / C5 / 16-prompt benches
O=gpurun_out/r06_final; mkdir -p $O
python -m pytest tests -m gpu -q -s > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1; tail -1 $O/bench_driver_cmd.log | cut -c1-400
bash tools/prof.sh r06_final/prof > $O/prof_head.txt 2>&1
db=$(find $O/prof -name "*.db" | head -1); python tools/forward_timeline.py $db > $O/forward_timeline.txt 2>&1; head -3 $O/forward_timeline.txt
cp $O/prof/kernel_stats.md $O/kernel_stats.md
(cd /tmp && export TMPDIR=/tmp && GILL_NO_GRAPH=1 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OLDPWD/$O/pmc -o m --output-format csv -- python $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --no-scale-origin > $OLDPWD/$O/pmc_run.log 2>&1)
python tools/pmc_mfma.py $O/pmc $O/pmc_mfma_busy.md | tail -3
python tools/pmc_by_kernel.py --infer-steps 4 --out $O/fetch_by_kernel.md | head -12
rm -rf $O/pmc $O/prof/prof
python bench.py --config c4 --steps 4 --warmup 2 --no-pmc > $O/bench_c4.log 2>&1; tail -n 1 $O/bench_c4.log | cut -c1-200
python bench.py --config c5 --steps 4 --warmup 2 --no-pmc > $O/bench_c5.log 2>&1; tail -n 1 $O/bench_c5.log | cut -c1-200
python bench.py --prompts-per-gpu 16 --steps 4 --warmup 2 --no-pmc --no-cpu-baseline > $O/bench_c2_16.log 2>&1; tail -n 1 $O/bench_c2_16.log | cut -c1-200
ls -la $O
