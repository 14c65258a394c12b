set +x
# GEMMs on 64 x 64-blocked weights above 256 rows on the general tiles (library) vs the STREAM64 tile at any M (tools/_lib_s64all.so = round 5)
O=gpurun_out/r06_s14; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm" > $O/ops_tests.log 2>&1; tail -2 $O/ops_tests.log
for P in 8 16 32; do for lib in tools/_lib_s64all.so gill_amd/libgill_amd.so; do echo -n "$lib  "; GILL_AMD_LIB=$(realpath $lib) python tools/opt_only.py $P 20 2>&1 | tail -1; done; done | tee $O/opt_ab.log
python -m pytest tests/test_coverage_gpu.py tests/test_stages_gpu.py -m gpu -q -x > $O/stage_tests.log 2>&1; tail -2 $O/stage_tests.log
