set +x
# UNet batch 16 (8 prompts per GPU: the multi-GPU configs' per-GPU load): the QKV ping-pong rule as built (admits the level-1 shape, 768 workgroups = three rounds) vs one round only
O=gpurun_out/r06_s33; mkdir -p $O
bash tools/ab_bench.sh tools/_lib_qkv1r.so gill_amd/libgill_amd.so 3 --prompts-per-gpu 8 2>&1 | tee $O/ab_loop.log
