set +x
# kernarg warm-up (common.h kernarg_warm): A/B against the same build without it (tools/_lib_kwoff.so), fixed cost by K sweep, quick parity
O=gpurun_out/r06_s09; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -m gpu -q -x > $O/ops_tests.log 2>&1; tail -2 $O/ops_tests.log
for lib in tools/_lib_kwoff.so gill_amd/libgill_amd.so; do echo "== $lib"; GILL_AMD_LIB=$(realpath $lib) python tools/conv_ksweep.py 2>&1 | tail -3; done | tee $O/ksweep.log
bash tools/ab_bench.sh tools/_lib_kwoff.so gill_amd/libgill_amd.so 3 2>&1 | tee $O/ab_loop.log
