set +x
O=gpurun_out/r06_s19; mkdir -p $O
for lib in gill_amd/libgill_amd.so tools/_lib_ppk.so; do echo "== $lib"; GILL_AMD_LIB=$(realpath $lib) python tools/pp_shortk_probe.py 2>&1 | grep GEMM; done | tee $O/probe.log
