set +x
O=gpurun_out/r06_s25; mkdir -p $O
bash tools/ab_bench.sh tools/_lib_base.so gill_amd/libgill_amd.so 5 2>&1 | tee $O/ab_loop.log
