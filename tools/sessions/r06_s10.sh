set +x
# timing-only: all prologue divisions as shifts (tools/_lib_pow2.so; exact on the power-of-two K-sweep shape) — the ceiling of trimming them
O=gpurun_out/r06_s10; mkdir -p $O
for r in 1 2 3; do for lib in gill_amd/libgill_amd.so tools/_lib_pow2.so; do echo "== $lib"; GILL_AMD_LIB=$(realpath $lib) python tools/conv_ksweep.py 2>&1 | tail -1; done; done | tee $O/ksweep.log
