set +x
# softmax-epilogue (cross-attention scores) GEMM on the 128 x 160 ping-pong tile where that is one balanced round: parity, loop A/B vs tools/_lib_base.so
O=gpurun_out/r06_s27; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "cross_attention or gemm" > $O/tests.log 2>&1; tail -2 $O/tests.log
bash tools/ab_bench.sh tools/_lib_base.so gill_amd/libgill_amd.so 4 2>&1 | tee $O/ab_loop.log
