set +x
O=gpurun_out/r06_s13; mkdir -p $O
GILL_AMD_LIB=$(realpath tools/_lib_wt.so) python -m pytest tests/test_ops_gpu.py -m gpu -q -x > $O/ops_tests_wt.log 2>&1; tail -2 $O/ops_tests_wt.log
bash tools/ab_bench.sh gill_amd/libgill_amd.so tools/_lib_wt.so 3 2>&1 | tee $O/ab_loop.log
