set +x
# conv_out_mfma_kernel with a row of taps in flight: parity (full-size UNet forwards, VAE), kernel time, loop A/B
O=gpurun_out/r06_s32; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "full_size or vae" > $O/tests.log 2>&1; tail -2 $O/tests.log
for lib in tools/_lib_base.so gill_amd/libgill_amd.so; do
  (cd /tmp && export TMPDIR=/tmp && GILL_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/p_$(basename $lib .so) -o b --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --no-scale-origin > $GRAFT_REPO_ROOT/$O/run_$(basename $lib .so).log 2>&1)
  f=$(find $O/p_$(basename $lib .so) -name "*kernel_stats.csv" | head -1)
  echo "== $lib"; grep -E "conv_out_mfma" $f | cut -c1-160
done | tee $O/conv_out.log
rm -rf $O/p_*
bash tools/ab_bench.sh tools/_lib_base.so gill_amd/libgill_amd.so 3 2>&1 | tee $O/ab_loop.log
