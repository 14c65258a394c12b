set +x
# QKV GEMMs on the ping-pong tiles (balanced rounds): parity (full-size forwards, the PP switch test through the loop), loop A/B vs tools/_lib_base.so, the QKV launches in the timeline
O=gpurun_out/r06_s24; mkdir -p $O
python -m pytest tests/test_configs_gpu.py tests/test_coverage_gpu.py -m gpu -q -x -k "full_size or pingpong or switch or batch or loop" > $O/tests.log 2>&1; tail -3 $O/tests.log
bash tools/ab_bench.sh tools/_lib_base.so gill_amd/libgill_amd.so 3 2>&1 | tee $O/ab_loop.log
bash tools/prof.sh r06_s24/prof > $O/prof_head.txt 2>&1
db=$(find $O/prof -name "*.db" | head -1); python tools/forward_timeline.py $db > $O/forward_timeline.txt 2>&1; grep -n "0, 3, " $O/forward_timeline.txt | head -14; tail -1 $O/forward_timeline.txt
rm -rf $O/prof/prof
