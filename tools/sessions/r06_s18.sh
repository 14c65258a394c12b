set +x
# interior-tile fast path of the row-major epilogue: parity, stamps, loop A/B against the previous build (tools/_lib_base.so)
O=gpurun_out/r06_s18; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -m gpu -q -x > $O/ops_tests.log 2>&1; tail -2 $O/ops_tests.log
GILL_AMD_LIB=$(realpath tools/_lib_stamps.so) python tools/stamps.py 2>&1 | grep -v amdgpu.ids > $O/stamps.log; grep -E "^==|partials published|last store issued|launch - last" $O/stamps.log
bash tools/ab_bench.sh tools/_lib_base.so gill_amd/libgill_amd.so 3 2>&1 | tee $O/ab_loop.log
