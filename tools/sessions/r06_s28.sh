set +x
# d = 160 attention on 64-query two-wave workgroups where 128-query ones leave half the CUs idle (UNet level 2): parity, loop A/B
O=gpurun_out/r06_s28; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "attention" > $O/tests.log 2>&1; tail -2 $O/tests.log
bash tools/ab_bench.sh tools/_lib_base.so gill_amd/libgill_amd.so 4 2>&1 | tee $O/ab_loop.log
