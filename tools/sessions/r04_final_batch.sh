set +x
mkdir -p gpurun_out/r04_final
python -m pytest tests -m gpu -q -s > gpurun_out/r04_final/gpu_tests.log 2>&1; tail -3 gpurun_out/r04_final/gpu_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_final/bench_driver_cmd.log 2>&1; tail -1 gpurun_out/r04_final/bench_driver_cmd.log | cut -c1-400
bash tools/prof.sh r04_final/prof > gpurun_out/r04_final/prof_head.txt 2>&1
db=$(find gpurun_out/r04_final/prof -name "*.db" | head -1); python tools/forward_timeline.py $db > gpurun_out/r04_final/forward_timeline.txt 2>&1; head -3 gpurun_out/r04_final/forward_timeline.txt
cp gpurun_out/r04_final/prof/kernel_stats.md gpurun_out/r04_final/kernel_stats.md
(cd /tmp && export TMPDIR=/tmp && GILL_NO_GRAPH=1 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OLDPWD/gpurun_out/r04_final/pmc -o m --output-format csv -- python $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --no-scale-origin > $OLDPWD/gpurun_out/r04_final/pmc_run.log 2>&1)
python tools/pmc_mfma.py gpurun_out/r04_final/pmc gpurun_out/r04_final/pmc_mfma_busy.md | tail -3
python tools/pmc_by_kernel.py --infer-steps 4 --out gpurun_out/r04_final/fetch_by_kernel.md | head -12
rm -rf gpurun_out/r04_final/pmc gpurun_out/r04_final/prof/prof
ls -la gpurun_out/r04_final
