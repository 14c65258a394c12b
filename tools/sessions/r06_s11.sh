set +x
# in-kernel stamps of the launch skeleton (tools/_lib_stamps.so, tools/stamps.py)
O=gpurun_out/r06_s11; mkdir -p $O
GILL_AMD_LIB=$(realpath tools/_lib_stamps.so) python tools/stamps.py 2>&1 | tee $O/stamps.log
