set +x
O=gpurun_out/r06_s31; mkdir -p $O
for lib in gill_amd/libgill_amd.so tools/_lib_tw160d3.so; do echo "== $lib"; GILL_AMD_LIB=$(realpath $lib) python tools/pp_shortk_probe.py 2>&1 | grep "2048 x 1280 x 1280\|512 x 1280"; done | tee $O/probe.log
bash tools/ab_bench.sh gill_amd/libgill_amd.so tools/_lib_tw160d3.so 3 2>&1 | tee $O/ab_loop.log
