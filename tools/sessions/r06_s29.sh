set +x
O=gpurun_out/r06_s29; mkdir -p $O
for lib in gill_amd/libgill_amd.so tools/_lib_npw1.so tools/_lib_npw2.so tools/_lib_npw4.so tools/_lib_npw5.so; do echo "== $lib"; GILL_AMD_LIB=$(realpath $lib) python tools/geglu_npw_probe.py 2>&1 | grep GEGLU; done | tee $O/probe.log
