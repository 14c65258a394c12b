set +x
# timing only: what the V^T 2-byte scatter of the QKV epilogue costs (tools/_lib_ablvt.so stores V like K: wrong values, same bytes, 16-B stores)
O=gpurun_out/r06_s16; mkdir -p $O
for lib in gill_amd/libgill_amd.so tools/_lib_ablvt.so; do
  (cd /tmp && export TMPDIR=/tmp && GILL_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/p_$(basename $lib .so) -o b --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --no-scale-origin > $GRAFT_REPO_ROOT/$O/run_$(basename $lib .so).log 2>&1)
  f=$(find $O/p_$(basename $lib .so) -name "*kernel_stats.csv" | head -1)
  echo "== $lib"; grep -E "gemm_kernel<8, 128, 0, 3|gemm_kernel<4, 128, 0, 3|attention_kernel<80|attention_kernel<160" $f | cut -c1-160
done | tee $O/qkv.log
rm -rf $O/p_*
