set +x
# write-through (sc1) stores of the GEMM / conv epilogues' finished bf16 outputs: stamps with and without, parity, loop A/B
O=gpurun_out/r06_s12; mkdir -p $O
for lib in tools/_lib_stamps.so tools/_lib_stampswt.so; do echo "######## $lib"; GILL_AMD_LIB=$(realpath $lib) python tools/stamps.py 2>&1 | grep -v amdgpu.ids; done | tee $O/stamps.log
GILL_AMD_LIB=$(realpath tools/_lib_wt.so) python -m pytest tests/test_ops_gpu.py -m gpu -q -x > $O/ops_tests_wt.log 2>&1; tail -2 $O/ops_tests_wt.log
bash tools/ab_bench.sh gill_amd/libgill_amd.so tools/_lib_wt.so 3 2>&1 | tee $O/ab_loop.log
