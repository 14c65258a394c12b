set +x
# plain GEMMs from K = 640 on the ping-pong tiles where their 128 x 160 tiling fills the chip: parity, timeline of the changed launches, loop A/B vs tools/_lib_base.so
O=gpurun_out/r06_s21; mkdir -p $O
python -m pytest tests/test_ops_gpu.py tests/test_configs_gpu.py -m gpu -q -x > $O/tests.log 2>&1; tail -2 $O/tests.log
bash tools/ab_bench.sh tools/_lib_base.so gill_amd/libgill_amd.so 3 2>&1 | tee $O/ab_loop.log
bash tools/prof.sh r06_s21/prof > $O/prof_head.txt 2>&1
db=$(find $O/prof -name "*.db" | head -1); python tools/forward_timeline.py $db > $O/forward_timeline.txt 2>&1; sed -n 24,34p $O/forward_timeline.txt; tail -1 $O/forward_timeline.txt
rm -rf $O/prof/prof
