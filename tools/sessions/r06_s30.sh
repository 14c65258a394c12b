set +x
# one N tile per workgroup where the multi-tile walk leaves the CUs unevenly loaded (level-2 GEGLU): parity, probe, loop A/B
O=gpurun_out/r06_s30; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "geglu or gemm" > $O/tests.log 2>&1; tail -2 $O/tests.log
for lib in tools/_lib_base.so gill_amd/libgill_amd.so; do echo "== $lib"; GILL_AMD_LIB=$(realpath $lib) python tools/geglu_npw_probe.py 2>&1 | grep GEGLU; done | tee $O/probe.log
bash tools/ab_bench.sh tools/_lib_base.so gill_amd/libgill_amd.so 4 2>&1 | tee $O/ab_loop.log
